"""Does LLM decode (side stream) overlap flow+HiFT (main stream)?  Times the phases alone and together."""
import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench as B
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
t0=time.perf_counter(); run_llm(); torch.cuda.synchronize(); t_llm=time.perf_counter()-t0
t0=time.perf_counter(); run_fh(); t_enq=time.perf_counter()-t0; torch.cuda.synchronize(); t_fh=time.perf_counter()-t0
side = torch.cuda.Stream(priority=-1)
t0=time.perf_counter(); run_fh(); t1=time.perf_counter()
with torch.cuda.stream(side): run_llm()
t2=time.perf_counter(); torch.cuda.synchronize(); t3=time.perf_counter()
print(f"llm alone {t_llm*1e3:.1f} ms | flow+hift alone {t_fh*1e3:.1f} ms (host enqueue {t_enq*1e3:.1f} ms) | together: enqueue {1e3*(t1-t0):.1f}, llm phase {1e3*(t2-t1):.1f}, total {1e3*(t3-t0):.1f} ms")
