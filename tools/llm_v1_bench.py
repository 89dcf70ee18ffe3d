"""CosyVoice-v1 TransformerLM decode rate at full size: K/V-cached decode steps vs full causal recompute per step."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cosyvoice_amd.config import TransformerLMConfig
from cosyvoice_amd.llm_v1 import TransformerLM
from cosyvoice_amd.weights import transformer_lm_state_dict

c = TransformerLMConfig.full()
m = TransformerLM(c, dtype=torch.float16, max_len=1024).load_state_dict(transformer_lm_state_dict(c, seed=3))
g = torch.Generator().manual_seed(4)
text = torch.randint(0, c.text_token_size, (1, 30), generator=g)
ptext = torch.randint(0, c.text_token_size, (1, 30), generator=g)
ps = torch.randint(0, c.speech_token_size, (1, 250), generator=g)      # 5 s prompt at 50 Hz
emb = torch.randn(1, c.spk_embed_dim, generator=g)
n = 250
forced = torch.randint(0, c.speech_token_size, (n,), generator=g).tolist()
for inc in (True, False):
    m.incremental = inc
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        toks = list(m._run(text, ptext, ps, emb, 0.0, 20.0, forced=forced))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"incremental={inc}: {len(toks)} tokens (context {1 + 1 + 60 + 1 + 250} -> +{n}) in {1e3 * dt:.1f} ms = {1e3 * dt / len(toks):.3f} ms/token, "
          f"{len(toks) / dt:.0f} tok/s = {len(toks) / dt / 50:.1f} x real time", flush=True)

# breakdown of one cached step: host enqueue vs GPU
m.incremental = True
list(m._run(text, ptext, ps, emb, 0.0, 20.0, forced=forced[:4]))
st = m.stack
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(100):
    st.decode_step(400 + i)
t_host = time.perf_counter() - t0
torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print(f"decode_step: host enqueue {1e3 * t_host / 100:.3f} ms, with GPU drain {1e3 * t_all / 100:.3f} ms per step ({len(st._dec.calls)} launches)", flush=True)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for i in range(100):
    st.decode_step(400 + i)
ev[1].record(); torch.cuda.synchronize()
print(f"decode_step GPU time {ev[0].elapsed_time(ev[1]) / 100:.3f} ms per step", flush=True)
