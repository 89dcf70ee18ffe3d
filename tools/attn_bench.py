import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from cosyvoice_amd import ops
B, H, T = 16, 8, 1000
dt = torch.float16; dev = 'cuda'
q = torch.randn(B, T, H * 64, device=dev).to(dt); k = torch.randn(B, T, H * 64, device=dev).to(dt)
Tp = 1000
vt = torch.randn(B, H, 64, Tp, device=dev).to(dt)
out = torch.zeros(B, T, H * 64, device=dev, dtype=dt)
def run():
    ops.attention(q, k, vt, out, B=B, H=H, Hkv=H, Tq=T, Tk=T, scale=0.125, q_bs=T * H * 64, ldq=H * 64, k_bs=T * H * 64, ldk=H * 64,
                  vt_ld=Tp, o_bs=T * H * 64, ldo=H * 64)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
e0.record()
for _ in range(n): run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / n * 1e3
print(f"attention B{B} H{H} T{T}: {us:.1f} us, {4.0*B*H*T*T*64/us/1e6:.1f} TF/s")
