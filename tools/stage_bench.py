"""Stage-level timings for BASELINE configs C2 (flow-matching decoder only: 10 Euler steps, 80 mel x 500 frames, one utterance)
and C3 (HiFT vocoder only: 22.05 kHz v1 generator, 10 s of mel = 861 frames), un-profiled, graphs warm."""
import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from cosyvoice_amd.config import FlowConfig, HiftConfig
from cosyvoice_amd.flow import CausalMaskedDiffWithXvec
from cosyvoice_amd.hift import HiFTGenerator
from cosyvoice_amd.weights import flow_state_dict, hift_state_dict

def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

fc = FlowConfig.full()
g = torch.Generator().manual_seed(0)
for dt in (torch.bfloat16, torch.float16):
    flow = CausalMaskedDiffWithXvec(fc, dtype=dt).load_state_dict(flow_state_dict(fc))
    flow.decoder.use_graph = True
    for n_p, n_g in ((0, 250), (250, 250)):
        tok = torch.randint(0, fc.vocab_size, (1, n_g), generator=g, dtype=torch.int32)
        ptok = torch.randint(0, fc.vocab_size, (1, n_p), generator=g, dtype=torch.int32)
        pfeat = torch.clamp(torch.randn(1, 2 * n_p, 80, generator=g) * 2 - 6, -11.5, 2.0)
        emb = torch.randn(1, fc.spk_embed_dim, generator=g)
        ms = timed(lambda: flow.inference_batch(tok, ptok, pfeat, emb))
        T = 2 * (n_p + n_g)
        print(f"C2 flow only [{dt}] prompt {2*n_p} + {2*n_g} frames (T={T}), encoder + 10 CFG Euler steps: {ms:.1f} ms "
              f"({493.7 * T / 1000 * 10 / ms:.0f} GFLOP/ms-scale: {0.4937 * T / 1000 * 10 / (ms * 1e-3):.0f} TFLOP/s on the estimator's {0.4937 * T / 1000 * 10:.2f} TFLOP)", flush=True)
    del flow
for tag, hc, T in (("v1 22.05 kHz", HiftConfig.v1(), 861), ("v2 24 kHz", HiftConfig.v2(), 500)):
    for mode in ("exact", "bf16x3"):
        hift = HiFTGenerator(hc, dtype=torch.float32, f32_products=mode).load_state_dict(hift_state_dict(hc))
        mel = torch.randn(1, 80, T, device="cuda")
        ms = timed(lambda: hift.inference(speech_feat=mel))
        gf = (518.37 if tag.startswith("v1") else 306.15)
        print(f"C3 HiFT only [{tag}, fp32 tensors, {mode} products] {T} frames -> {T * hc.total_upsample} samples: {ms:.2f} ms "
              f"({gf / ms:.1f} TFLOP/s of {gf:.0f} GFLOP; RTF {ms * 1e-3 / (T * hc.total_upsample / hc.sampling_rate):.5f})", flush=True)
