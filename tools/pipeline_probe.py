"""Steady-state balance of the partitioned pipeline: two decode loops on k CU slots per XCD, flow + HiFT on the rest — each
side alone and both together (who finishes when)."""
import sys, time, threading, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench as B
from cosyvoice_amd import ops
from cosyvoice_amd.config import FlowConfig, HiftConfig, LlmConfig
from cosyvoice_amd.flow import CausalMaskedDiffWithXvec
from cosyvoice_amd.hift import HiFTGenerator
from cosyvoice_amd.llm import Qwen2LM
from cosyvoice_amd.weights import flow_state_dict, hift_state_dict, llm_state_dict

lc, fc, hc = LlmConfig.full(), FlowConfig.full(), HiftConfig.v2()
llm = Qwen2LM(lc, dtype=torch.bfloat16, max_batch=8, ctx_max=704, max_out=258); llm.load_state_dict(llm_state_dict(lc))
llms = [llm, llm.new_context(), llm.new_context()]
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
for m in llms: run_llm(m)
run_fh(); torch.cuda.synchronize()
for k, nl in ((8, 2), (8, 3), (12, 2)):
    sl = [ops.masked_stream(lambda s, x: s < k) for _ in range(nl)]
    sf = ops.masked_stream(lambda s, x: s >= k)
    for m, s in zip(llms, sl):
        with torch.cuda.stream(s): run_llm(m)
    with torch.cuda.stream(sf): run_fh()
    torch.cuda.synchronize()
    def timed_parallel(with_llm, with_flow, reps_flow=2):
        done = {}
        def lw(i):
            with torch.no_grad(), torch.cuda.stream(sl[i]):
                run_llm(llms[i]); torch.cuda.current_stream().synchronize(); done[f"llm{i}"] = time.perf_counter() - t0
        def fw():
            with torch.no_grad(), torch.cuda.stream(sf):
                for r in range(reps_flow):
                    run_fh(); torch.cuda.current_stream().synchronize(); done[f"flow{r}"] = time.perf_counter() - t0
        ths = ([threading.Thread(target=lw, args=(i,)) for i in range(nl)] if with_llm else []) + ([threading.Thread(target=fw)] if with_flow else [])
        t0 = time.perf_counter()
        for t in ths: t.start()
        for t in ths: t.join()
        return {k_: round(v * 1e3, 1) for k_, v in sorted(done.items())}
    print(f"k={k} loops={nl}: LLM loops alone {timed_parallel(True, False)} | flow+HiFT x2 alone {timed_parallel(False, True)} | together {timed_parallel(True, True)}", flush=True)
