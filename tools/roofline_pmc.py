"""Runs ONLY the roofline kernel of bench.py (decode gate/up skinny GEMM, B rows = argv[1], default 16 = two 8-utterance batches
in one token loop; all 24 layers' packed weights, 5 sweeps) so that rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes give its HBM
traffic per launch (tools/pmc_summarize.py turns the two counter CSVs into profiles/r01_roofline_pmc*.json)."""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from cosyvoice_amd import ops
H, I, L = 896, 4864, 24
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = 'cuda'
torch.manual_seed(0)
R = 16 if B <= 16 else 32
xn = torch.zeros(R, H, device=dev, dtype=torch.bfloat16); xn[:B] = torch.randn(B, H, device=dev).to(torch.bfloat16)
x = torch.randn(R, H, device=dev); gam = torch.ones(H, device=dev)
h = torch.zeros(R, I, device=dev, dtype=torch.bfloat16)
packs = [ops.pack_skinny((torch.randn(2 * I, H, device=dev) / H ** 0.5).to(torch.bfloat16), interleave=True) for _ in range(L)]
split = len(sys.argv) > 2 and sys.argv[2] == "split"      # the decode step's form: 16-bit rows + partial sums, 1/rms in the epilogue
ssp = torch.rand(56, R, device=dev)
for _ in range(5):
    for p in packs:
        if split:
            ops.skinny_gemm(xn, p, B, 2 * I, H, mode=2, out_act=h, ldoa=I, split_in=dict(rs=ssp, n=56, eps=1e-6))
        else:
            ops.skinny_gemm(xn, p, B, 2 * I, H, mode=2, out_act=h, ldoa=I, norm=dict(x=x, gamma=gam, eps=1e-6))
torch.cuda.synchronize()
