"""Is the direct-launch host path a bottleneck?  Mean host time of one decode-step replay (124 hipLaunchKernel calls inside
cv_graph_launch_direct) with the two decode loops alone and with the flow thread replaying its graphs beside them."""
import sys, time, threading, collections, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench as B
from cosyvoice_amd import ops
from cosyvoice_amd.config import FlowConfig, HiftConfig, LlmConfig
from cosyvoice_amd.flow import CausalMaskedDiffWithXvec
from cosyvoice_amd.hift import HiFTGenerator
from cosyvoice_amd.llm import Qwen2LM
from cosyvoice_amd.weights import flow_state_dict, hift_state_dict, llm_state_dict

acc = collections.defaultdict(lambda: [0.0, 0])
_orig = ops.Graph.launch
def timed_launch(self):
    t0 = time.perf_counter(); _orig(self); dt = time.perf_counter() - t0
    a = acc[threading.current_thread().name]; a[0] += dt; a[1] += 1
ops.Graph.launch = timed_launch

lc, fc, hc = LlmConfig.full(), FlowConfig.full(), HiftConfig.v2()
llm = Qwen2LM(lc, dtype=torch.bfloat16, max_batch=8, ctx_max=704, max_out=258); llm.load_state_dict(llm_state_dict(lc))
llms = [llm, llm.new_context()]
flow = CausalMaskedDiffWithXvec(fc, dtype=torch.float16); flow.load_state_dict(flow_state_dict(fc))
hift = HiFTGenerator(hc, dtype=torch.float32); hift.load_state_dict(hift_state_dict(hc))
flow.decoder.use_graph = True
texts, forced, ptext, pspeech, pfeat, emb = B.make_inputs(lc, fc, 100)
dev = 'cuda'; Bn = 8
texts_d = [t.to(dev) for t in texts]; pt, ps, pf, em = ptext.to(dev), pspeech.to(dev), pfeat.to(dev), emb.to(dev)
tok = torch.tensor(forced, dtype=torch.int32, device=dev)
def run_llm(m): return m.generate_batch(texts_d, [pt]*Bn, [ps]*Bn, forced=forced, steps_per_poll=64)
def run_fh():
    mel = flow.inference_batch(tok, ps.expand(Bn,-1), pf.expand(Bn,-1,-1), em.expand(Bn,-1))
    return hift.inference(speech_feat=mel.contiguous(), cache_source=torch.zeros(1,1,0))[0]
k = 8
sl = [ops.masked_stream(lambda s, x: s < k) for _ in range(2)]
sf = ops.masked_stream(lambda s, x: s >= k)
for m, s in zip(llms, sl):
    with torch.cuda.stream(s): run_llm(m)
with torch.cuda.stream(sf): run_fh()
torch.cuda.synchronize()
def go(with_flow):
    acc.clear()
    def lw(i):
        with torch.no_grad(), torch.cuda.stream(sl[i]): run_llm(llms[i])
    def fw():
        with torch.no_grad(), torch.cuda.stream(sf):
            run_fh(); run_fh()
    ths = [threading.Thread(target=lw, args=(i,), name=f"llm{i}") for i in range(2)] + ([threading.Thread(target=fw, name="flow")] if with_flow else [])
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths[:2]: t.join()
    t_llm = time.perf_counter() - t0
    for t in ths[2:]: t.join()
    torch.cuda.synchronize()
    return t_llm, {n: (round(1e6 * a[0] / a[1], 1), a[1]) for n, a in sorted(acc.items())}
for wf in (False, True):
    t, a = go(wf)
    print(f"flow thread {'on ' if wf else 'off'}: decode loops done after {1e3*t:.1f} ms; host us per Graph.launch (mean, calls): {a}", flush=True)
