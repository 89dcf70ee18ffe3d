"""GPU: CosyVoice-v1 TransformerLM against the reference-minted golden (teacher-forced log-probabilities
of the reference's own cached decode loop) and the oracle; the reference-signature generator."""
import os

import numpy as np
import pytest
import torch

from cosyvoice_amd.config import TransformerLMConfig
from cosyvoice_amd.weights import transformer_lm_state_dict

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("incremental", [True, False])
@pytest.mark.parametrize("dt,tol", [(torch.float16, 2e-2), (torch.bfloat16, 1.5e-1)])
def test_forced_logp_vs_reference_golden(golden_dir, dt, tol, incremental):
    """incremental: prefill + K/V-cached decode steps (the product path); not incremental: full causal recompute per step."""
    from cosyvoice_amd.llm_v1 import TransformerLM
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, "llm_v1_tiny.npz")).items()}
    c = TransformerLMConfig.tiny()
    m = TransformerLM(c, dtype=dt, max_len=256).load_state_dict(transformer_lm_state_dict(c, seed=int(g["seed"])))
    m.incremental = incremental
    lp = m.forced_logp(g["text"], g["prompt_text"], g["prompt_speech_token"], g["embedding"], g["forced"].tolist())
    ref = g["logp"]
    assert lp.shape == ref.shape
    fin = torch.isfinite(ref)
    assert (~fin).sum() == 1 and torch.isinf(lp[0, c.speech_token_size])        # EOS masked at the first step (llm.py:227-229)
    err = (lp[fin] - ref[fin]).abs().max().item()
    agree = (lp[:, :c.speech_token_size].argmax(-1) == ref[:, :c.speech_token_size].argmax(-1)).float().mean().item()
    print(f"v1 llm logp[{dt}, incremental={incremental}] Linf {err:.3e}, argmax agreement {agree:.2f}")
    assert err < tol and agree >= 0.9


def test_lm_input_and_generator_vs_oracle():
    from cosyvoice_amd.llm_v1 import TransformerLM
    from oracle import llm_v1 as o1
    c = TransformerLMConfig.tiny()
    sd = transformer_lm_state_dict(c, seed=7)
    m = TransformerLM(c, dtype=torch.float16, max_len=512).load_state_dict(sd)
    g = torch.Generator().manual_seed(2)
    for L, Lp, N, spk in ((5, 2, 6, True), (3, 0, 0, False)):
        text = torch.randint(0, c.text_token_size, (1, L), generator=g)
        ptext = torch.randint(0, c.text_token_size, (1, Lp), generator=g)
        ps = torch.randint(0, c.speech_token_size, (1, N), generator=g)
        emb = torch.randn(1, c.spk_embed_dim, generator=g) if spk else torch.zeros(0, c.spk_embed_dim)
        ref = o1.lm_input(sd, c, text, ptext, ps, emb)[0]
        x = m.lm_input(text, ptext, ps, emb).cpu()
        assert x.shape == ref.shape and (x - ref).abs().max().item() < 2e-2
        toks = list(m.inference(text=text, text_len=torch.tensor([L], dtype=torch.int32), prompt_text=ptext,
                                prompt_text_len=torch.tensor([Lp], dtype=torch.int32), prompt_speech_token=ps,
                                prompt_speech_token_len=torch.tensor([N], dtype=torch.int32), embedding=emb))
        assert 2 * L <= len(toks) <= 20 * L and all(isinstance(t, int) and 0 <= t < c.speech_token_size for t in toks)


def test_cached_decode_matches_recompute_full_size():
    """FULL-size stack (14 layers, 1024 d, 16 heads): the K/V-cached decode path against the full causal recompute on the same
    teacher-forced ids, across a bucket boundary of the recompute path (prompt 50 rows, 30 steps)."""
    from cosyvoice_amd.llm_v1 import TransformerLM
    c = TransformerLMConfig.full()
    m = TransformerLM(c, dtype=torch.float16, max_len=512).load_state_dict(transformer_lm_state_dict(c, seed=3))
    g = torch.Generator().manual_seed(4)
    text = torch.randint(0, c.text_token_size, (1, 20), generator=g)
    ptext = torch.randint(0, c.text_token_size, (1, 8), generator=g)
    ps = torch.randint(0, c.speech_token_size, (1, 19), generator=g)
    emb = torch.randn(1, c.spk_embed_dim, generator=g)
    forced = torch.randint(0, c.speech_token_size, (30,), generator=g).tolist()
    m.incremental = True
    a = m.forced_logp(text, ptext, ps, emb, forced)
    assert m.n_decode_steps == len(forced)             # one prefill, then one cached step per emitted token
    m.incremental = False
    b = m.forced_logp(text, ptext, ps, emb, forced)
    assert m.n_decode_steps == len(forced)
    fin = torch.isfinite(b)
    err = (a[fin] - b[fin]).abs().max().item()
    agree = (a[:, :-1].argmax(-1) == b[:, :-1].argmax(-1)).float().mean().item()
    print(f"v1 llm FULL size: cached vs recompute Linf {err:.3e}, argmax agreement {agree:.2f}")
    assert err < 2e-2 and agree >= 0.95
