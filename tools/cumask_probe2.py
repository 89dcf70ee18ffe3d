"""Does giving the LLM decode graph its own CUs fix the LLM || flow overlap?  hipGraph replays ignore a stream's CU
mask (cumask_probe.py), direct launches honour it: here flow+HiFT run as direct launches (use_graph=False) on a stream
masked to a subset of CUs from a second host thread, while the LLM graph replays run unmasked at high priority."""
import ctypes as C, sys, time, threading, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench as B
from cosyvoice_amd.config import FlowConfig, HiftConfig, LlmConfig
from cosyvoice_amd.flow import CausalMaskedDiffWithXvec
from cosyvoice_amd.hift import HiFTGenerator
from cosyvoice_amd.llm import Qwen2LM
from cosyvoice_amd.model import CosyVoice2Model
from cosyvoice_amd.weights import flow_state_dict, hift_state_dict, llm_state_dict
hip = C.CDLL("libamdhip64.so")


def masked_stream(keep, layout):
    """keep(cu_slot 0..31, xcd 0..7) -> bool.  layout 'mod': mask bit i <-> (xcd i%8, slot i//8); 'div': (xcd i//32, slot i%32)."""
    words = (C.c_uint32 * 8)()
    n = 0
    for i in range(256):
        xcd, slot = (i % 8, i // 8) if layout == 'mod' else (i // 32, i % 32)
        if keep(slot, xcd):
            words[i // 32] |= (1 << (i % 32)); n += 1
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value), n


lc, fc, hc = LlmConfig.full(), FlowConfig.full(), HiftConfig.v2()
llm = Qwen2LM(lc, dtype=torch.bfloat16, max_batch=8, ctx_max=704, max_out=258)
flow = CausalMaskedDiffWithXvec(fc, dtype=torch.float16); hift = HiFTGenerator(hc, dtype=torch.float32)
model = CosyVoice2Model(llm, flow, hift).load_state_dicts(llm_state_dict(lc), flow_state_dict(fc), hift_state_dict(hc))
texts, forced, ptext, pspeech, pfeat, emb = B.make_inputs(lc, fc, 100)
dev = 'cuda'; Bn = 8
texts_d = [t.to(dev) for t in texts]; pt, ps, pf, em = ptext.to(dev), pspeech.to(dev), pfeat.to(dev), emb.to(dev)
tok = torch.tensor(forced, dtype=torch.int32, device=dev)
def run_llm(): return llm.generate_batch(texts_d, [pt]*Bn, [ps]*Bn, forced=forced)
def run_fh():
    mel = flow.inference_batch(tok, ps.expand(Bn,-1), pf.expand(Bn,-1,-1), em.expand(Bn,-1))
    return hift.inference(speech_feat=mel.contiguous(), cache_source=torch.zeros(1,1,0))[0]

flow.decoder.use_graph = True
for _ in range(2): run_llm(); run_fh()
torch.cuda.synchronize()
t0 = time.perf_counter(); run_llm(); torch.cuda.synchronize(); print(f"LLM graph alone: {1e3*(time.perf_counter()-t0):.1f} ms", flush=True)
t0 = time.perf_counter(); run_fh(); torch.cuda.synchronize(); print(f"flow graph alone: {1e3*(time.perf_counter()-t0):.1f} ms", flush=True)
flow.decoder.use_graph = False
run_fh(); torch.cuda.synchronize()
t0 = time.perf_counter(); run_fh(); te = time.perf_counter() - t0; torch.cuda.synchronize()
print(f"flow direct alone: {1e3*(time.perf_counter()-t0):.1f} ms (host enqueue {1e3*te:.1f})", flush=True)

hi = torch.cuda.Stream(priority=-1)
cases = [("full", lambda s, x: True, 'mod')]
for layout in ('mod', 'div'):
    cases += [(f"24/32 per XCD [{layout}]", lambda s, x: s < 24, layout), (f"28/32 per XCD [{layout}]", lambda s, x: s < 28, layout),
              (f"7 of 8 XCDs [{layout}]", lambda s, x: x < 7, layout), (f"6 of 8 XCDs [{layout}]", lambda s, x: x < 6, layout)]
for name, keep, layout in cases:
    sf, n = masked_stream(keep, layout)
    with torch.cuda.stream(sf):
        run_fh(); torch.cuda.synchronize()
        t0 = time.perf_counter(); run_fh(); torch.cuda.synchronize(); t_alone = time.perf_counter() - t0
    def worker():
        with torch.cuda.stream(sf):
            run_fh()
    t0 = time.perf_counter()
    th = threading.Thread(target=worker); th.start()
    with torch.cuda.stream(hi):
        run_llm()
    t_llm = time.perf_counter() - t0
    th.join(); torch.cuda.synchronize(); t = time.perf_counter() - t0
    print(f"flow on {n} CUs ({name}): alone {1e3*t_alone:.1f} ms | together {1e3*t:.1f} ms (LLM phase {1e3*t_llm:.1f})", flush=True)
