"""Micro-benchmark of cv_gemm on the flow/HiFT/prefill shapes (events on the launch stream, interleaved rounds)."""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from cosyvoice_amd import ops

def bench(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us

dev = 'cuda'
shapes = [("qk", 16000, 1024, 256), ("ff1", 16000, 1024, 256), ("ff2", 16000, 256, 1024), ("out", 16000, 256, 512),
          ("conv3", 16000, 256, 768), ("prefill_gu", 2256, 9728, 896), ("enc_ff1", 8000, 2048, 512)]
for dt in (torch.float16, torch.float32):
    for name, M, N, K in shapes:
        if dt == torch.float32 and M > 8000: continue
        x = torch.randn(M, K, device=dev).to(dt); W = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt)
        bias = torch.randn(N, device=dev); res = torch.randn(M, N, device=dev)
        o32 = torch.empty(M, N, device=dev); oa = torch.empty(M, N, device=dev, dtype=dt)
        t_plain = bench(lambda: ops.linear(x, W, out_act=oa))
        t_full = bench(lambda: ops.linear(x, W, bias=bias, res=res, out_f32=o32, out_act=oa, act=ops.ACT_GELU))
        t_ref = bench(lambda: torch.matmul(x, W.t()))
        fl = 2.0 * M * N * K
        print(f"{str(dt)[6:]:8s} {name:11s} M{M} N{N} K{K}: plain {t_plain:7.1f} us ({fl/t_plain/1e6:6.1f} TF/s)  bias+res+gelu+2out {t_full:7.1f} us  torch.matmul {t_ref:7.1f} us ({fl/t_ref/1e6:6.1f} TF/s)")
