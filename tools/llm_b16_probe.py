"""Decode step cost at batch 16 vs batch 8 (one loop), all CUs and on 64 CUs: does sharing the weight stream between two
utterance batches beat two separate decode loops?"""
import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench as B
from cosyvoice_amd import ops
from cosyvoice_amd.config import FlowConfig, LlmConfig
from cosyvoice_amd.llm import Qwen2LM
from cosyvoice_amd.weights import llm_state_dict
lc, fc = LlmConfig.full(), FlowConfig.full()
llm = Qwen2LM(lc, dtype=torch.bfloat16, max_batch=16, ctx_max=704, max_out=258)
llm.load_state_dict(llm_state_dict(lc))
texts, forced, ptext, pspeech, pfeat, emb = B.make_inputs(lc, fc, 100)
dev = 'cuda'
pt, ps = ptext.to(dev), pspeech.to(dev)
def run(Bn, steps=None):
    tx = [texts[i % 8].to(dev) for i in range(Bn)]
    fo = [forced[i % 8] for i in range(Bn)]
    return llm.generate_batch(tx, [pt]*Bn, [ps]*Bn, forced=fo, steps_per_poll=64, max_steps=steps)
def timed(stream, Bn, steps=None):
    best = 1e9
    with torch.cuda.stream(stream):
        run(Bn, steps); torch.cuda.synchronize()
        for _ in range(2):
            t0 = time.perf_counter(); run(Bn, steps); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return 1e3 * best
for name, s in (("all CUs", torch.cuda.current_stream()), ("64 CUs", ops.masked_stream(lambda slot, x: slot < 8)), ("32 CUs", ops.masked_stream(lambda slot, x: slot < 4))):
    for Bn in (8, 16):
        tp, ta = timed(s, Bn, 1), timed(s, Bn)
        print(f"{name} B={Bn}: prefill+1 {tp:.1f} ms, full {ta:.1f} ms -> {1e3*(ta-tp)/249:.1f} us/step", flush=True)
