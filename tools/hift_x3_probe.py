"""HiFT decoder convs: exact-f32 MFMA vs bf16x3 split products — error vs the reference goldens and time (B=8, T=500)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from cosyvoice_amd.config import HiftConfig
from cosyvoice_amd.hift import HiFTGenerator
from cosyvoice_amd.weights import hift_state_dict
for tag, cfg in (("hift_tiny", HiftConfig.tiny()), ("hift_v2", HiftConfig.v2()), ("hift_v1", HiftConfig.v1())):
    gd = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(gd, tag + ".npz")).items()}
    print(tag, list(g.keys())[:8])
    for mode in ("exact", "bf16x3"):
        m = HiFTGenerator(cfg, dtype=torch.float32, f32_products=mode).load_state_dict(hift_state_dict(cfg))
        wav = m.decode(g["mel"].cuda(), g["s"].cuda())
        if wav is not None:
            print(tag, mode, "decode err", (wav.cpu() - g["wav"]).abs().max().item())
cfg = HiftConfig.v2()
for mode in ("exact", "bf16x3"):
    m = HiFTGenerator(cfg, dtype=torch.float32, f32_products=mode).load_state_dict(hift_state_dict(cfg))
    mel = torch.randn(8, 80, 500, device="cuda")
    for _ in range(2): m.inference(speech_feat=mel)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): m.inference(speech_feat=mel)
    torch.cuda.synchronize(); print(mode, "HiFT B8 T500:", (time.perf_counter() - t0) / 5 * 1e3, "ms")
