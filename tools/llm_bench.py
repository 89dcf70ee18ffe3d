"""LLM stage alone (C4 shape: batch 8, prefill 282, 250 forced tokens): all CUs (hipGraph replays) and on the CU share the
pipeline gives it (direct launches on a masked stream).  Usage: python tools/llm_bench.py [slots_per_xcd ...]"""
import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench as B
from cosyvoice_amd import ops
from cosyvoice_amd.config import FlowConfig, LlmConfig
from cosyvoice_amd.llm import Qwen2LM
from cosyvoice_amd.weights import llm_state_dict

lc, fc = LlmConfig.full(), FlowConfig.full()
llm = Qwen2LM(lc, dtype=torch.bfloat16, max_batch=8, ctx_max=704, max_out=258)
llm.load_state_dict(llm_state_dict(lc))
texts, forced, ptext, pspeech, pfeat, emb = B.make_inputs(lc, fc, 100)
dev = 'cuda'; Bn = 8
texts_d = [t.to(dev) for t in texts]; pt, ps = ptext.to(dev), pspeech.to(dev)
def run(steps=None): return llm.generate_batch(texts_d, [pt]*Bn, [ps]*Bn, forced=forced, steps_per_poll=64, max_steps=steps)
def timed(stream, steps=None, reps=3):
    best = 1e9
    with torch.cuda.stream(stream):
        run(steps); torch.cuda.synchronize()
        for _ in range(reps):
            t0 = time.perf_counter(); run(steps); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return 1e3 * best
cur = torch.cuda.current_stream()
t_pre, t_all = timed(cur, 1), timed(cur)
print(f"all CUs (graph): prefill+1 step {t_pre:.1f} ms, full {t_all:.1f} ms -> {1e3*(t_all-t_pre)/249:.1f} us/step", flush=True)
for k in [int(a) for a in sys.argv[1:]] or [12]:
    s = ops.masked_stream(lambda slot, x: slot < k)
    llm.cu_budget = 0
    t_pre, t_all = timed(s, 1), timed(s)
    print(f"{8*k} CUs (direct): prefill+1 step {t_pre:.1f} ms, full {t_all:.1f} ms -> {1e3*(t_all-t_pre)/249:.1f} us/step", flush=True)
