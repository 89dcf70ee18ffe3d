import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from cosyvoice_amd import ops
M, N, K = 16000, 1024, 256
dev='cuda'; dt=torch.float16
x = torch.randn(M, K, device=dev).to(dt); W = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt)
oa = torch.empty(M, N, device=dev, dtype=dt)
for _ in range(5): ops.linear(x, W, out_act=oa)
torch.cuda.synchronize()
