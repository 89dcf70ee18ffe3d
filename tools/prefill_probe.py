"""Prefill (+ first token) wall time of Qwen2LM for NB utterances (env NB, default 8) and a forced GEMM tile (env CV_GEMM_TILE): what the
decode jobs of tts_batches put on the flow CUs per job.  B = 8: 6.7 ms, B = 32: 18.8 ms (16.2 with the 8-wave 128x128 tile)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from cosyvoice_amd.config import FlowConfig, LlmConfig
from cosyvoice_amd.llm import Qwen2LM
from cosyvoice_amd.weights import llm_state_dict
lc, fc = LlmConfig.full(), FlowConfig.full()
nb = int(os.environ.get("NB", "8"))
llm = Qwen2LM(lc, dtype=torch.bfloat16, max_batch=nb, ctx_max=704, max_out=258).load_state_dict(llm_state_dict(lc))
texts, forced, ptext, pspeech, pfeat, emb = B.make_inputs(lc, fc, 100, n_utts=nb)
dev = "cuda"
t = [x.to(dev) for x in texts]; pt, ps = ptext.to(dev), pspeech.to(dev)
def run(): llm.generate_batch(t, [pt] * nb, [ps] * nb, forced=forced, max_steps=1)
for _ in range(3): run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): run()
torch.cuda.synchronize(); print(f"prefill(+1 step) B={nb} tile={os.environ.get('CV_GEMM_TILE')}: {(time.perf_counter()-t0)/5*1e3:.2f} ms")
