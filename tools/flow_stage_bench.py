"""The bench's flow + HiFT stage alone (batch 8, 10 s prompt + 10 s generated: T = 1000, encoder + 10 CFG Euler steps captured as
one hipGraph, then the vocoder) on all CUs, un-profiled: ms per batch and TFLOP/s on the estimator's 4.937 TF per utterance
(SURVEY.md §8d).  CV_FLOW_FUSED=0 times the unfused cv_gemm / cv_layernorm launches for comparison."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from cosyvoice_amd.config import FlowConfig, HiftConfig, LlmConfig
from cosyvoice_amd.flow import CausalMaskedDiffWithXvec
from cosyvoice_amd.hift import HiFTGenerator
from cosyvoice_amd.weights import flow_state_dict, hift_state_dict

dt = torch.float16 if (len(sys.argv) < 2 or sys.argv[1] == "fp16") else torch.bfloat16
Bn = int(sys.argv[2]) if len(sys.argv) > 2 else 8
lc, fc, hc = LlmConfig.full(), FlowConfig.full(), HiftConfig.v2()
flow = CausalMaskedDiffWithXvec(fc, dtype=dt).load_state_dict(flow_state_dict(fc))
flow.decoder.use_graph = True
hift = HiFTGenerator(hc, dtype=torch.float32).load_state_dict(hift_state_dict(hc))
texts, forced, ptext, pspeech, pfeat, emb = B.make_inputs(lc, fc, 100)
dev = "cuda"
tok = torch.tensor(forced, dtype=torch.int32, device=dev)[:Bn]
args = (tok, pspeech.to(dev).expand(Bn, -1), pfeat.to(dev).expand(Bn, -1, -1), emb.to(dev).expand(Bn, -1))
zero = torch.zeros(1, 1, 0)

def timed(fn, n=8):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

mel = flow.inference_batch(*args).clone()
ms_f = timed(lambda: flow.inference_batch(*args))
ms_h = timed(lambda: hift.inference(speech_feat=mel.contiguous(), cache_source=zero))
tf = 4.937 * Bn
print(f"flow stage [{dt}, fused={flow.decoder.estimator.fused}] batch {Bn}, T=1000: flow {ms_f:.1f} ms ({tf / ms_f * 1e3:.0f} TFLOP/s on {tf:.1f} TF), "
      f"HiFT {ms_h:.1f} ms, sum {ms_f + ms_h:.1f} ms; mel absmean {mel.abs().mean().item():.4f}", flush=True)
