"""First-packet latency of CosyVoice2Model.tts(stream=True) at the C4 shape (10 s prompt, free-running decode capped by forced
tokens): time from the call to the first yielded chunk, chunk cadence, total."""
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
llm = Qwen2LM(lc, dtype=torch.bfloat16, max_batch=1, ctx_max=1024, max_out=600)
flow = CausalMaskedDiffWithXvec(fc, dtype=torch.float16); hift = HiFTGenerator(hc, dtype=torch.float32)
model = CosyVoice2Model(llm, flow, hift).load_state_dicts(llm_state_dict(lc), flow_state_dict(fc), hift_state_dict(hc))
flow.decoder.use_graph = True
texts, forced, ptext, pspeech0, pfeat0, emb = B.make_inputs(lc, fc, 100)
bucket = int(sys.argv[1]) if len(sys.argv) > 1 else 25
model.stream_length_bucket = bucket
print(f"stream_length_bucket = {bucket}")
for rep, cut in enumerate((0, 0, 7, 13, 7)):          # prompt shortened by `cut` tokens: a request of a new length
    pspeech, pfeat = pspeech0[:, :250 - cut], pfeat0[:, :2 * (250 - cut)]
    t0 = time.perf_counter()
    stamps, samples = [], []
    for out in model.tts(text=texts[0], flow_embedding=emb, llm_embedding=torch.zeros(0, 192), prompt_text=ptext,
                         llm_prompt_speech_token=pspeech, flow_prompt_speech_token=pspeech, prompt_speech_feat=pfeat, stream=True):
        stamps.append(time.perf_counter() - t0)
        samples.append(out["tts_speech"].shape[1])
    audio = sum(samples) / 24000
    gaps = [b - a for a, b in zip(stamps, stamps[1:])]
    print(f"run {rep} (prompt {pspeech.shape[1]} tokens): first chunk after {1e3*stamps[0]:.0f} ms ({samples[0]/24000:.2f} s of audio), {len(stamps)} chunks, "
          f"mean gap {1e3*sum(gaps)/max(len(gaps),1):.0f} ms, total {1e3*stamps[-1]:.0f} ms for {audio:.1f} s of audio (RTF {stamps[-1]/audio:.3f})", flush=True)
