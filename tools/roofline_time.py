"""Un-profiled time of bench.py's roofline kernel (decode gate/up skinny GEMM + RMSNorm prologue + SwiGLU) at 8 and 16 rows:
the 24 layers' launches captured once, replayed between events (what bench.measure_roofline does, without the pipeline)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cosyvoice_amd import ops
H, I, L = 896, 4864, 24
dev = "cuda"
torch.manual_seed(0)
packs = [ops.pack_skinny((torch.randn(2 * I, H, device=dev) / H ** 0.5).to(torch.bfloat16), interleave=True) for _ in range(L)]
xn = torch.zeros(16, H, device=dev, dtype=torch.bfloat16)
x = torch.randn(16, H, device=dev); gam = torch.ones(H, device=dev)
h = torch.zeros(16, I, device=dev, dtype=torch.bfloat16)
for B in (8, 16):
    g = ops.Graph().capture(lambda: [ops.skinny_gemm(xn, p, B, 2 * I, H, mode=2, out_act=h, ldoa=I, norm=dict(x=x, gamma=gam, eps=1e-6)) for p in packs])
    g.launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        g.launch()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (50 * L)
    alg = 2 * I * H * 2 + B * H * 4 + H * 4 + B * I * 2
    print(f"rows {B}: {us:.2f} us per launch, {alg / us / 1e6:.2f} TB/s = {alg / us / 8e6:.3f} of 8 TB/s", flush=True)
# the same launches without the fused RMSNorm prologue (A read as 16-bit rows): what the prologue costs at each row count
for B in (8, 16):
    g = ops.Graph().capture(lambda: [ops.skinny_gemm(xn, p, B, 2 * I, H, mode=2, out_act=h, ldoa=I) for p in packs])
    g.launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        g.launch()
    e1.record(); torch.cuda.synchronize()
    print(f"rows {B}, no norm prologue: {e0.elapsed_time(e1) * 1e3 / (50 * L):.2f} us per launch", flush=True)

# split-RMSNorm form (what the decode step runs): 16-bit rows + per-workgroup partial sums from the producer, 1/rms in the epilogue
ssp = torch.rand(56, 16, device=dev)
for B in (8, 16):
    g = ops.Graph().capture(lambda: [ops.skinny_gemm(xn, p, B, 2 * I, H, mode=2, out_act=h, ldoa=I, split_in=dict(rs=ssp, n=56, eps=1e-6)) for p in packs])
    g.launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        g.launch()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (50 * L)
    alg = 2 * I * H * 2 + B * H * 2 + 56 * 16 * 4 + B * I * 2
    print(f"rows {B}, split norm: {us:.2f} us per launch, {alg / us / 1e6:.2f} TB/s = {alg / us / 8e6:.3f} of 8 TB/s", flush=True)
