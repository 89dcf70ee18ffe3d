"""One batch-8 pass of the flow (encoder + 10 CFG Euler steps, T = 1000, eager launches) and HiFT — the workload of the bench's
flow + vocoder stage — for `rocprofv3 --pmc ...` passes (MFMA busy cycles, LDS bank conflicts, wave / wait cycles per kernel).
tools/pmc_kernels.py aggregates the counter CSV per kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from cosyvoice_amd.config import FlowConfig, HiftConfig, LlmConfig
from cosyvoice_amd.flow import CausalMaskedDiffWithXvec
from cosyvoice_amd.hift import HiFTGenerator
from cosyvoice_amd.weights import flow_state_dict, hift_state_dict

lc, fc, hc = LlmConfig.full(), FlowConfig.full(), HiftConfig.v2()
flow = CausalMaskedDiffWithXvec(fc, dtype=torch.float16).load_state_dict(flow_state_dict(fc))
hift = HiFTGenerator(hc, dtype=torch.float32).load_state_dict(hift_state_dict(hc))
texts, forced, ptext, pspeech, pfeat, emb = B.make_inputs(lc, fc, 100)
dev, Bn = "cuda", 8
tok = torch.tensor(forced, dtype=torch.int32, device=dev)
for _ in range(2):     # first pass builds workspaces / tables, the second is the one to read
    mel = flow.inference_batch(tok, pspeech.to(dev).expand(Bn, -1), pfeat.to(dev).expand(Bn, -1, -1), emb.to(dev).expand(Bn, -1))
    wav, _ = hift.inference(speech_feat=mel.contiguous(), cache_source=torch.zeros(1, 1, 0))
torch.cuda.synchronize()
print("flow + hift pass done", tuple(wav.shape))
