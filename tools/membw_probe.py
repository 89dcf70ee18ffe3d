"""Write / copy bandwidth at the activation working-set sizes of the flow stage (16 .. 64 MB buffers, reused back to back, so
the Infinity Cache is in play): what a kernel that writes its 48 MB of [Q | K | V^T] can expect."""
import torch, time
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for mb in (16, 48, 64, 128, 512):
    n = mb * 1024 * 1024 // 2
    a = torch.empty(n, dtype=torch.float16, device="cuda"); b = torch.empty_like(a)
    us_w = t(lambda: a.fill_(1.0)); us_c = t(lambda: b.copy_(a))
    bufs = [torch.empty(n, dtype=torch.float16, device="cuda") for _ in range(max(1, 1024 // mb))]
    i = [0]
    def rot():
        i[0] = (i[0] + 1) % len(bufs); bufs[i[0]].fill_(1.0)
    us_r = t(rot)
    print(f"{mb:4d} MB: fill {us_w:7.1f} us = {mb*1.048576/us_w*1e0:6.2f} TB/s | copy {us_c:7.1f} us = {2*mb*1.048576/us_c:6.2f} TB/s (r+w) | fill rotating over 1 GB {us_r:7.1f} us = {mb*1.048576/us_r:6.2f} TB/s", flush=True)
