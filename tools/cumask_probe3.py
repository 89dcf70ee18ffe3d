"""LLM decode and flow+HiFT on DISJOINT CU sets: both replayed launch by launch (cv_graph_launch_direct) on CU-masked
streams, the flow side from a second host thread.  k = CU slots per XCD given to the LLM."""
import sys, time, threading, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench as B
from cosyvoice_amd import ops
from cosyvoice_amd.config import FlowConfig, HiftConfig, LlmConfig
from cosyvoice_amd.flow import CausalMaskedDiffWithXvec
from cosyvoice_amd.hift import HiFTGenerator
from cosyvoice_amd.llm import Qwen2LM
from cosyvoice_amd.model import CosyVoice2Model
from cosyvoice_amd.weights import flow_state_dict, hift_state_dict, llm_state_dict

lc, fc, hc = LlmConfig.full(), FlowConfig.full(), HiftConfig.v2()
llm = Qwen2LM(lc, dtype=torch.bfloat16, max_batch=8, ctx_max=704, max_out=258)
flow = CausalMaskedDiffWithXvec(fc, dtype=torch.float16); hift = HiFTGenerator(hc, dtype=torch.float32)
model = CosyVoice2Model(llm, flow, hift).load_state_dicts(llm_state_dict(lc), flow_state_dict(fc), hift_state_dict(hc))
texts, forced, ptext, pspeech, pfeat, emb = B.make_inputs(lc, fc, 100)
dev = 'cuda'; Bn = 8
texts_d = [t.to(dev) for t in texts]; pt, ps, pf, em = ptext.to(dev), pspeech.to(dev), pfeat.to(dev), emb.to(dev)
tok = torch.tensor(forced, dtype=torch.int32, device=dev)
def run_llm(): return llm.generate_batch(texts_d, [pt]*Bn, [ps]*Bn, forced=forced, steps_per_poll=64)
def run_fh():
    mel = flow.inference_batch(tok, ps.expand(Bn,-1), pf.expand(Bn,-1,-1), em.expand(Bn,-1))
    return hift.inference(speech_feat=mel.contiguous(), cache_source=torch.zeros(1,1,0))[0]
flow.decoder.use_graph = True
for _ in range(2): run_llm(); run_fh()
torch.cuda.synchronize()
def timed(fn, stream=None):
    with torch.cuda.stream(stream or torch.cuda.current_stream()):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0)
print(f"graph replays, unmasked: LLM {timed(run_llm):.1f} ms, flow+HiFT {timed(run_fh):.1f} ms", flush=True)
full = ops.masked_stream(lambda s, x: True)
print(f"direct launches, all CUs: LLM {timed(run_llm, full):.1f} ms, flow+HiFT {timed(run_fh, full):.1f} ms", flush=True)
for k in (10, 12, 14):
    sl = ops.masked_stream(lambda s, x: s < k)
    sf = ops.masked_stream(lambda s, x: s >= k)
    tl, tf = timed(run_llm, sl), timed(run_fh, sf)
    def worker():
        with torch.cuda.stream(sf):
            run_fh()
    t0 = time.perf_counter()
    th = threading.Thread(target=worker); th.start()
    with torch.cuda.stream(sl):
        run_llm()
    t_llm = time.perf_counter() - t0
    th.join(); torch.cuda.synchronize(); t = time.perf_counter() - t0
    print(f"LLM on {8*k} CUs alone {tl:.1f} ms | flow+HiFT on {256-8*k} CUs alone {tf:.1f} ms | together {1e3*t:.1f} ms (LLM done at {1e3*t_llm:.1f})", flush=True)
