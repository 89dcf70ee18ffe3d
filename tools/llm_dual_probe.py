"""Two independent decode loops (two utterance batches, own KV caches / state) on the SAME CU-masked partition, from two host
threads: does interleaving two latency-bound dependency chains raise the decode stage's throughput?"""
import sys, time, threading, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench as B
from cosyvoice_amd import ops
from cosyvoice_amd.config import FlowConfig, LlmConfig
from cosyvoice_amd.llm import Qwen2LM
from cosyvoice_amd.weights import llm_state_dict

lc, fc = LlmConfig.full(), FlowConfig.full()
sd = llm_state_dict(lc)
llms = []
for _ in range(3):
    m = Qwen2LM(lc, dtype=torch.bfloat16, max_batch=8, ctx_max=704, max_out=258)
    m.load_state_dict(sd)
    llms.append(m)
texts, forced, ptext, pspeech, pfeat, emb = B.make_inputs(lc, fc, 100)
dev = 'cuda'; Bn = 8
texts_d = [t.to(dev) for t in texts]; pt, ps = ptext.to(dev), pspeech.to(dev)
def run(m): return m.generate_batch(texts_d, [pt]*Bn, [ps]*Bn, forced=forced, steps_per_poll=64)
for m in llms: run(m)
torch.cuda.synchronize()
for k in (8, 10, 12):
    streams = [ops.masked_stream(lambda slot, x: slot < k) for _ in range(3)]
    for m, s in zip(llms, streams):
        with torch.cuda.stream(s):
            run(m)
    torch.cuda.synchronize()
    res = []
    for n in (1, 2, 3):
        def worker(i):
            with torch.no_grad(), torch.cuda.stream(streams[i]): run(llms[i])
        t0 = time.perf_counter()
        ths = [threading.Thread(target=worker, args=(i,)) for i in range(n)]
        for th in ths: th.start()
        for th in ths: th.join()
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) * 1e3)
    print(f"{8*k} CUs: 1 loop {res[0]:.1f} ms | 2 loops {res[1]:.1f} = {res[1]/2:.1f} per batch | 3 loops {res[2]:.1f} = {res[2]/3:.1f} per batch", flush=True)
