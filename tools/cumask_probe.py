"""Does a CU-masked stream (hipExtStreamCreateWithCUMask) constrain hipGraph replays?  LLM decode alone on masks of
256 / 128 / 64 CUs, then flow+HiFT on the complementary mask concurrently."""
import ctypes as C, sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench as B
from cosyvoice_amd.config import FlowConfig, HiftConfig, LlmConfig
from cosyvoice_amd.flow import CausalMaskedDiffWithXvec
from cosyvoice_amd.hift import HiFTGenerator
from cosyvoice_amd.llm import Qwen2LM
from cosyvoice_amd.model import CosyVoice2Model
from cosyvoice_amd.weights import flow_state_dict, hift_state_dict, llm_state_dict
hip = C.CDLL("libamdhip64.so")
def masked_stream(xcd_bits):
    # 256 CUs = 8 x 32-bit words; CU numbering interleaves XCDs in the mask on MI300-class parts: bit i -> XCD i%8.
    words = (C.c_uint32 * 8)()
    for cu in range(256):
        if (xcd_bits >> (cu % 8)) & 1: words[cu // 32] |= (1 << (cu % 32))
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)
lc, fc, hc = LlmConfig.full(), FlowConfig.full(), HiftConfig.v2()
llm = Qwen2LM(lc, dtype=torch.bfloat16, max_batch=8, ctx_max=704, max_out=258)
flow = CausalMaskedDiffWithXvec(fc, dtype=torch.float16); hift = HiFTGenerator(hc, dtype=torch.float32)
model = CosyVoice2Model(llm, flow, hift).load_state_dicts(llm_state_dict(lc), flow_state_dict(fc), hift_state_dict(hc))
flow.decoder.use_graph = True
texts, forced, ptext, pspeech, pfeat, emb = B.make_inputs(lc, fc, 100)
dev = 'cuda'; Bn = 8
texts_d = [t.to(dev) for t in texts]; pt, ps, pf, em = ptext.to(dev), pspeech.to(dev), pfeat.to(dev), emb.to(dev)
tok = torch.tensor(forced, dtype=torch.int32, device=dev)
def run_llm(): return llm.generate_batch(texts_d, [pt]*Bn, [ps]*Bn, forced=forced)
def run_fh():
    mel = flow.inference_batch(tok, ps.expand(Bn,-1), pf.expand(Bn,-1,-1), em.expand(Bn,-1))
    return hift.inference(speech_feat=mel.contiguous(), cache_source=torch.zeros(1,1,0))[0]
for _ in range(2): run_llm(); run_fh()
torch.cuda.synchronize()
for bits, name in ((0xFF, "8 XCD"), (0x0F, "4 XCD"), (0x03, "2 XCD"), (0x01, "1 XCD")):
    s = masked_stream(bits)
    with torch.cuda.stream(s):
        run_llm(); torch.cuda.synchronize()
        t0 = time.perf_counter(); run_llm(); torch.cuda.synchronize(); t = time.perf_counter() - t0
    print(f"LLM alone on {name}: {t*1e3:.1f} ms")
for lbits, fbits, name in ((0x03, 0xFC, "LLM 2 XCD | flow 6 XCD"), (0x01, 0xFE, "LLM 1 | flow 7"), (0x0F, 0xF0, "LLM 4 | flow 4")):
    sl, sf = masked_stream(lbits), masked_stream(fbits)
    with torch.cuda.stream(sf): run_fh()
    with torch.cuda.stream(sl): run_llm()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(sf): run_fh()
    with torch.cuda.stream(sl): run_llm()
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    print(f"together [{name}]: {t*1e3:.1f} ms")
