"""Flow + HiFT stage on the pipeline's CU share (slots >= k of every XCD) for different numbers of utterances per pass: ms per pass and
per utterance.  The row-block kernels run in whole rounds of workgroups over the CUs of the stream, so the cost per utterance is a
saw-tooth in the batch size.   python tools/flow_share_probe.py [k=8] [B ...]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from cosyvoice_amd import ops
from cosyvoice_amd.config import FlowConfig, HiftConfig, LlmConfig
from cosyvoice_amd.flow import CausalMaskedDiffWithXvec
from cosyvoice_amd.hift import HiFTGenerator
from cosyvoice_amd.weights import flow_state_dict, hift_state_dict

k = int(sys.argv[1]) if len(sys.argv) > 1 else 8
Bs = [int(a) for a in sys.argv[2:]] or [4, 5, 6, 7, 8, 9, 10, 12]
lc, fc, hc = LlmConfig.full(), FlowConfig.full(), HiftConfig.v2()
flow = CausalMaskedDiffWithXvec(fc, dtype=torch.float16).load_state_dict(flow_state_dict(fc))
flow.decoder.use_graph = True
hift = HiFTGenerator(hc, dtype=torch.float32).load_state_dict(hift_state_dict(hc))
texts, forced, ptext, pspeech, pfeat, emb = B.make_inputs(lc, fc, 100, n_utts=max(Bs))
dev = "cuda"
zero = torch.zeros(1, 1, 0)
st = ops.masked_stream(lambda s, x: s >= k) if k > 0 else torch.cuda.current_stream()
est = flow.decoder.estimator
# PROBE_DECODE=rows: two decode loops of that many rows run on the other CUs meanwhile (what the flow stage sees inside the pipeline)
dec_rows = int(os.environ.get("PROBE_DECODE", "0"))
stop = False
if dec_rows and k > 0:
    import threading
    from cosyvoice_amd.llm import Qwen2LM
    from cosyvoice_amd.weights import llm_state_dict
    llm = Qwen2LM(lc, dtype=torch.float16, max_batch=dec_rows, ctx_max=704, max_out=B.N_GEN + 8).load_state_dict(llm_state_dict(lc))
    ctxs = [llm, llm.new_context()]
    t_d, f_d, pt_d, ps_d, _, _ = B.make_inputs(lc, fc, 100, n_utts=dec_rows)
    t_d = [t.to(dev) for t in t_d]
    steps_done = [0, 0]

    def loop(i):
        sd = ops.masked_stream(lambda s, x: s < k)
        with torch.no_grad(), torch.cuda.stream(sd):
            while not stop:
                ctxs[i].generate_batch(t_d, [pt_d.to(dev)] * dec_rows, [ps_d.to(dev)] * dec_rows, forced=f_d, steps_per_poll=64)
                steps_done[i] += 1
    ths = [threading.Thread(target=loop, args=(i,), daemon=True) for i in range(2)]
    for t in ths:
        t.start()
    time.sleep(3.0)   # both loops past their first (capturing) job
for Bn in Bs:
    tok = torch.tensor(forced[:Bn], dtype=torch.int32, device=dev)
    args = (tok, pspeech.to(dev).expand(Bn, -1), pfeat.to(dev).expand(Bn, -1, -1), emb.to(dev).expand(Bn, -1))
    est.cu_budget = (32 - k) * 8 if k > 0 else 0
    with torch.no_grad(), torch.cuda.stream(st):
        def run():
            mel = flow.inference_batch(*args)
            return hift.inference(speech_feat=mel.contiguous(), cache_source=zero)[0]
        def run_flow():
            return flow.inference_batch(*args)
        res = []
        for fn in (run_flow, run):
            for _ in range(2):
                fn()
            st.synchronize()
            t0 = time.perf_counter()
            n = 4
            for _ in range(n):
                fn()
            st.synchronize()
            res.append((time.perf_counter() - t0) / n * 1e3)
    print(f"k={k} ({(32 - k) * 8 if k else 256} CUs){' + 2 decode loops x %d rows' % dec_rows if dec_rows else ''} B={Bn:2d}: flow {res[0]:6.1f} ms = {res[0] / Bn:5.2f} ms/utt | flow + HiFT {res[1]:6.1f} ms = {res[1] / Bn:5.2f} ms/utt", flush=True)
stop = True
if dec_rows and k > 0:
    for t in ths:
        t.join(timeout=30)
    print(f"decode jobs finished meanwhile: {steps_done}")
