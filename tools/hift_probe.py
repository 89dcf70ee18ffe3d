import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from cosyvoice_amd.config import HiftConfig
from cosyvoice_amd.weights import hift_state_dict
from cosyvoice_amd.hift import HiFTGenerator
from oracle import hift as oh
cfg = HiftConfig.v2(); sd = hift_state_dict(cfg)
torch.manual_seed(0)
T = 100
mel = torch.clamp(torch.randn(1, 80, T) * 2 - 6, -11.5, 2.0)
ph, nz = oh.draw_source_randoms(cfg, 1, T * cfg.total_upsample, seed=5)
wav_ref, s_ref = oh.inference(sd, cfg, mel, None, ph, nz, scan_dtype=torch.float64)
mel8 = torch.clamp(torch.randn(8, 80, 500) * 2 - 6, -11.5, 2.0).cuda()
for dt in (torch.float32, torch.float16, torch.bfloat16):
    m = HiFTGenerator(cfg, dtype=dt).load_state_dict(sd)
    wav, s = m.inference(mel.cuda(), torch.zeros(1, 1, 0), ph.cuda(), nz.cuda())
    err = (wav.cpu() - wav_ref).abs()
    for _ in range(2): m.inference(mel8)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): m.inference(mel8)
    torch.cuda.synchronize(); dtm = (time.perf_counter() - t0) / 3
    print(f"{dt}: wav Linf {err.max().item():.3e} mean {err.mean().item():.3e} (ref absmax {wav_ref.abs().max().item():.3f}); 8x500 frames {dtm*1e3:.1f} ms")
