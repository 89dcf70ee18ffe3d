"""HiFT decode alone, for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/pmc_hbm.py -> profiles/r03_hift_hbm.json) and for
--kernel-trace --stats: N decodes of one utterance, fp32 tensors with bf16x3 products (the bench's setting).
argv: v2|v1 (24 kHz 500 frames | 22.05 kHz 861 frames: BASELINE C3), N (default 4; the first decode also builds the workspaces), B (utterances, default 1).
Weight-side kernels (packing at load) run before the decodes and are part of the totals: the tool prints the names to subtract."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cosyvoice_amd.config import HiftConfig
from cosyvoice_amd.hift import HiFTGenerator
from cosyvoice_amd.weights import hift_state_dict

tag = sys.argv[1] if len(sys.argv) > 1 else "v2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1     # utterances per decode
cfg, frames = (HiftConfig.v2(), 500) if tag == "v2" else (HiftConfig.v1(), 861)
m = HiFTGenerator(cfg, dtype=torch.float32, f32_products=os.environ.get("HIFT_PRODUCTS", "bf16x3")).load_state_dict(hift_state_dict(cfg))
g = torch.Generator().manual_seed(0)
mel = torch.clamp(torch.randn(B, 80, frames, generator=g) * 2 - 6, -11.5, 2.0).cuda()
s = (torch.randn(B, 1, frames * cfg.total_upsample, generator=g) * 0.05).cuda()
torch.cuda.synchronize()
import time
w = m.decode(mel, s)          # builds the workspaces
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n - 1):
    w = m.decode(mel, s)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / max(n - 1, 1) * 1e3
print(f"hift {tag}: {n} decodes of {B} x {frames} frames -> {w.shape[1]} samples, absmax {w.abs().max().item():.3f}; {ms:.3f} ms per decode (host clock)")
