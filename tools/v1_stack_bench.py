"""CosyVoice-v1 stack at full size on one MI355X: TransformerLM (cached decode) -> MaskedDiffWithXvec -> HiFT v1 (22.05 kHz).
Stage times for a 10 s utterance behind a 3 s prompt, and the end-to-end CosyVoiceModel.tts() real-time factor."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cosyvoice_amd.config import FlowV1Config, HiftConfig, TransformerLMConfig
from cosyvoice_amd.flow_v1 import MaskedDiffWithXvec
from cosyvoice_amd.hift import HiFTGenerator
from cosyvoice_amd.llm_v1 import TransformerLM
from cosyvoice_amd.model import CosyVoiceModel
from cosyvoice_amd.weights import flow_v1_state_dict, hift_state_dict, transformer_lm_state_dict

lc, fc, hc = TransformerLMConfig.full(), FlowV1Config.full(), HiftConfig.v1()
llm = TransformerLM(lc, dtype=torch.float16, max_len=2048).load_state_dict(transformer_lm_state_dict(lc))
flow = MaskedDiffWithXvec(fc, dtype=torch.float16).load_state_dict(flow_v1_state_dict(fc))
hift = HiFTGenerator(hc, dtype=torch.float32).load_state_dict(hift_state_dict(hc))
m = CosyVoiceModel(llm, flow, hift, fp16=False, sr=22050)
g = torch.Generator().manual_seed(0)
n_p, n_g, sr = 150, 500, 22050
ptok = torch.randint(0, fc.vocab_size, (1, n_p), generator=g, dtype=torch.int32)
tok = torch.randint(0, fc.vocab_size, (1, n_g), generator=g, dtype=torch.int32)
t1 = flow.mel_len(n_p, sr)
pfeat = torch.clamp(torch.randn(1, t1, 80, generator=g) * 2 - 6, -11.5, 2.0)
emb = torch.randn(1, fc.spk_embed_dim, generator=g)
kw = dict(token=tok, token_len=torch.tensor([n_g]), prompt_token=ptok, prompt_token_len=torch.tensor([n_p]), prompt_feat=pfeat,
          prompt_feat_len=torch.tensor([t1]), embedding=emb, flow_cache=torch.zeros(1, 80, 0, 2), sample_rate=sr)


def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, out


for graph in (False, True):
    flow.decoder.use_graph = graph
    flow.inference(**kw); flow.inference(**kw)
    dt, (mel, _) = timed(lambda: flow.inference(**kw))
    print(f"flow v1 (T = {t1} + {mel.shape[2]} frames, 10 Euler steps x 2 CFG rows, graph={graph}): {1e3 * dt:.1f} ms", flush=True)
dt, (wav, _) = timed(lambda: hift.inference(speech_feat=mel, cache_source=torch.zeros(1, 1, 0)))
secs = wav.shape[1] / sr
print(f"HiFT v1 ({mel.shape[2]} frames -> {wav.shape[1]} samples = {secs:.2f} s): {1e3 * dt:.1f} ms", flush=True)

# end to end: 50 text tokens; the LM decides the length (synthetic weights: random length between 2x and 20x the text)
text = torch.randint(0, lc.text_token_size, (1, 50), generator=g)
ptext = torch.randint(0, lc.text_token_size, (1, 20), generator=g)
lemb = torch.randn(1, lc.spk_embed_dim, generator=g)
args = dict(text=text, flow_embedding=emb, llm_embedding=lemb, prompt_text=ptext, llm_prompt_speech_token=ptok,
            flow_prompt_speech_token=ptok, prompt_speech_feat=pfeat)
for stream in (False, True):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter(); first = None; n = 0
        for o in m.tts(stream=stream, **args):
            if first is None:
                torch.cuda.synchronize(); first = time.perf_counter() - t0
            n += o["tts_speech"].shape[1]
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"tts(stream={stream}): {n / sr:.2f} s of audio in {1e3 * dt:.0f} ms (RTF {dt / (n / sr):.4f}), first chunk after {1e3 * first:.0f} ms", flush=True)
