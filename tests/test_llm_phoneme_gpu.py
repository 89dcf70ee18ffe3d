"""GPU: Qwen2LM_Phoneme_Src2 (the fork's production LM) — prefill front-end against the reference-minted golden / the oracle,
the non_random_ras_sampling mode of the sampler kernel against the oracle, and the reference-signature generator."""
import os

import numpy as np
import pytest
import torch

from cosyvoice_amd.config import LlmConfig, PhonemeFrontConfig
from cosyvoice_amd.weights import phoneme_lm_state_dict

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dt,tol", [(torch.float16, 2e-2), (torch.bfloat16, 1.5e-1)])
def test_lm_input_vs_reference_golden(golden_dir, dt, tol):
    from cosyvoice_amd.llm_phoneme import Qwen2LM_Phoneme_Src2
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, "llm_phoneme_tiny.npz")).items()}
    lc, pc = LlmConfig.tiny(), PhonemeFrontConfig.tiny()
    sd = phoneme_lm_state_dict(pc, lc, seed=int(g["seed"]))
    m = Qwen2LM_Phoneme_Src2(lc, pc, dtype=dt, max_batch=4, ctx_max=256, max_out=256).load_state_dict(sd)
    x = m.lm_input(g["text"], g["pho"], g["prompt_text"], g["prompt_pho"], g["prompt_speech_token"], g["embedding"]).cpu()
    ref = g["lm_input"][0]
    assert x.shape == ref.shape
    err = (x - ref).abs()
    print(f"lm_input[{dt}]: Linf {err.max().item():.3e} L1 {err.mean().item():.3e} (|ref| max {ref.abs().max().item():.2f})")
    assert err.max().item() < tol and err.mean().item() < tol / 8
    # rows that are pure table look-ups (sos, task id, prompt speech) are exact
    P_ = g["pho"].shape[1] + g["prompt_pho"].shape[1]
    assert torch.equal(x[0], ref[0]) and torch.equal(x[2 + P_], ref[2 + P_]) and torch.equal(x[3 + P_:], ref[3 + P_:])


def test_lm_input_vs_oracle_no_speaker_no_prompt():
    from cosyvoice_amd.llm_phoneme import Qwen2LM_Phoneme_Src2
    from oracle import llm_phoneme as op
    lc, pc = LlmConfig.tiny(), PhonemeFrontConfig.tiny()
    sd = phoneme_lm_state_dict(pc, lc, seed=3)
    m = Qwen2LM_Phoneme_Src2(lc, pc, dtype=torch.float16, max_batch=4, ctx_max=256, max_out=256).load_state_dict(sd)
    g = torch.Generator().manual_seed(2)
    for L, P_, Lp, Pp, N, spk in ((5, 9, 0, 0, 0, False), (1, 1, 0, 0, 4, True), (8, 70, 2, 3, 0, True)):
        pho = torch.stack([torch.randint(0, n, (1, P_), generator=g) for n in (pc.text_token_size, pc.text_tone_size, pc.text_lang_size,
                                                                               pc.text_prsd_size)], dim=-1)
        ppho = torch.stack([torch.randint(0, n, (1, Pp), generator=g) for n in (pc.text_token_size, pc.text_tone_size, pc.text_lang_size,
                                                                                pc.text_prsd_size)], dim=-1)
        text = torch.randint(0, lc.vocab_size, (1, L), generator=g)
        ptext = torch.randint(0, lc.vocab_size, (1, Lp), generator=g)
        ps = torch.randint(0, lc.speech_token_size, (1, N), generator=g)
        emb = torch.randn(1, pc.spk_embed_dim, generator=g) if spk else torch.zeros(0, pc.spk_embed_dim)
        ref = op.phoneme_lm_input(sd, pc, lc, text, pho, ptext, ppho, ps, emb)[0]
        x = m.lm_input(text, pho, ptext, ppho, ps, emb).cpu()
        assert x.shape == ref.shape == (1 + int(spk) + P_ + Pp + 1 + N, lc.hidden_size)
        assert (x - ref).abs().max().item() < 2e-2


def test_sampler_wider_nucleus_fallback_matches_oracle():
    """fallback_mode 1 = non_random_ras_sampling (utils/common.py:116-123): on a repetition the second draw is a nucleus with
    (top_p + 0.15, top_k * 2) — token-exact against the oracle on identical logits with injected uniforms."""
    from cosyvoice_amd import _lib as L
    from cosyvoice_amd import ops
    from oracle import llm_phoneme as op
    dev = "cuda"
    V, eos, B, H = 6564, 6561, 8, 64
    g = torch.Generator().manual_seed(19)
    logits = torch.randn(B, 6576, generator=g) * 3.0
    hist_len = 6
    hist = torch.randint(0, 6561, (B, hist_len), generator=g, dtype=torch.int32)
    top1 = logits[:, :V].argmax(-1)
    for b in (1, 2, 6):                      # force the repetition branch
        hist[b, -1 - (b % 3)] = top1[b]
        logits[b, top1[b]] += 6.0
    uni = torch.rand(B, 101, 2, generator=g) * 0.98
    emb = torch.randn(V, H, generator=g)
    st = dict(step=torch.full((B,), 30, dtype=torch.int32), pos=torch.full((B,), 50, dtype=torch.int32),
              n_emitted=torch.full((B,), hist_len, dtype=torch.int32), finished=torch.zeros(B, dtype=torch.int32),
              min_len=torch.zeros(B, dtype=torch.int32), max_len=torch.full((B,), 100, dtype=torch.int32))
    out_tokens = torch.zeros(B, 32, dtype=torch.int32)
    out_tokens[:, :hist_len] = hist
    d = {k: v.to(dev) for k, v in st.items()}
    lg_d, uni_d, emb_d, out_d = logits.to(dev), uni.to(dev), emb.to(dev), out_tokens.to(dev)
    x_d = torch.zeros(B, H, device=dev)
    p = L.SampleParams()
    p.logits, p.ldl, p.V, p.B = lg_d.data_ptr(), 6576, V, B
    p.eos, p.top_k, p.top_p, p.win_size, p.tau_r = eos, 10, 0.8, 10, 0.1
    p.fallback_mode, p.top_p2, p.top_k2 = 1, 0.95, 20
    p.seed, p.uniforms, p.max_trials = 0, uni_d.data_ptr(), 100
    p.min_len, p.max_len = d["min_len"].data_ptr(), d["max_len"].data_ptr()
    p.forced, p.forced_ld = None, 0
    p.step, p.pos, p.n_emitted, p.finished = d["step"].data_ptr(), d["pos"].data_ptr(), d["n_emitted"].data_ptr(), d["finished"].data_ptr()
    p.out_tokens, p.out_ld = out_d.data_ptr(), 32
    p.emb_table, p.emb_dim, p.x, p.ldx = emb_d.data_ptr(), H, x_d.data_ptr(), H
    ops.sample_ras(p)
    torch.cuda.synchronize()
    ne, fin, toks = d["n_emitted"].cpu(), d["finished"].cpu(), out_d.cpu()
    n_rep = 0
    for b in range(B):
        lp = logits[b, :V].log_softmax(-1)
        ref = op.non_random_ras_sampling(lp, hist[b].tolist(), tuple(uni[b, 0].tolist()))
        first = op.ol.nucleus_sampling(lp, uni[b, 0, 0].item(), 0.8, 10)
        n_rep += int(first in hist[b].tolist()[-10:])
        if ref == eos:
            assert int(fin[b]) == 1
        elif ref > eos:
            assert int(ne[b]) == hist_len
        else:
            assert int(ne[b]) == hist_len + 1 and int(toks[b, hist_len]) == ref, (b, ref, int(toks[b, hist_len]))
    assert n_rep >= 3


def test_reference_signature_generator():
    from cosyvoice_amd.llm_phoneme import Qwen2LM_Phoneme_Src2
    lc, pc = LlmConfig.tiny(), PhonemeFrontConfig.tiny()
    sd = phoneme_lm_state_dict(pc, lc, seed=5, round_to=torch.bfloat16)
    m = Qwen2LM_Phoneme_Src2(lc, pc, dtype=torch.bfloat16, max_batch=4, ctx_max=256, max_out=256).load_state_dict(sd)
    g = torch.Generator().manual_seed(4)
    L, P_ = 5, 12
    pho = torch.stack([torch.randint(0, n, (1, P_), generator=g) for n in (pc.text_token_size, pc.text_tone_size, pc.text_lang_size,
                                                                           pc.text_prsd_size)], dim=-1)
    text = torch.randint(0, lc.vocab_size, (1, L), generator=g)
    e_i, e_p = torch.zeros(1, 0, dtype=torch.int64), torch.zeros(1, 0, 4, dtype=torch.int64)
    toks = list(m.inference(text=(text, pho), text_len=(torch.tensor([L], dtype=torch.int32), torch.tensor([P_], dtype=torch.int32)),
                            prompt_text=(e_i, e_p), prompt_text_len=(torch.tensor([0], dtype=torch.int32), torch.tensor([0], dtype=torch.int32)),
                            prompt_speech_token=e_i, prompt_speech_token_len=torch.tensor([0], dtype=torch.int32),
                            embedding=torch.randn(1, pc.spk_embed_dim, generator=g)))
    assert 2 * L <= len(toks) <= 20 * L and all(isinstance(t, int) and 0 <= t < lc.speech_token_size for t in toks)


def test_lm_input_reference_dimensions_vs_oracle():
    """The production dimensions of the front-end (phoneme factors 400+64+16+32, 6 x 1024-wide conformer layers with 16 heads,
    DecoderLayer over hidden 896 = 16 heads x 56 channels, zero-padded to the kernel's 64-wide heads, FFN 4096) on a 1-layer
    Qwen2 of hidden size 896, against the oracle."""
    import dataclasses
    from cosyvoice_amd.llm_phoneme import Qwen2LM_Phoneme_Src2
    from oracle import llm_phoneme as op
    lc = dataclasses.replace(LlmConfig.tiny(), hidden_size=896, num_heads=14, num_kv_heads=2, num_layers=1)
    pc = PhonemeFrontConfig.full()
    sd = phoneme_lm_state_dict(pc, lc, seed=8)
    m = Qwen2LM_Phoneme_Src2(lc, pc, dtype=torch.float16, max_batch=2, ctx_max=256, max_out=64).load_state_dict(sd)
    g = torch.Generator().manual_seed(6)
    L, P_, N = 12, 41, 9
    pho = torch.stack([torch.randint(0, n, (1, P_), generator=g) for n in (pc.text_token_size, pc.text_tone_size, pc.text_lang_size,
                                                                           pc.text_prsd_size)], dim=-1)
    text = torch.randint(0, lc.vocab_size, (1, L), generator=g)
    ps = torch.randint(0, lc.speech_token_size, (1, N), generator=g)
    emb = torch.randn(1, pc.spk_embed_dim, generator=g)
    e_i, e_p = torch.zeros(1, 0, dtype=torch.int64), torch.zeros(1, 0, 4, dtype=torch.int64)
    ref = op.phoneme_lm_input(sd, pc, lc, text, pho, e_i, e_p, ps, emb)[0]
    x = m.lm_input(text, pho, e_i, e_p, ps, emb).cpu()
    err = (x - ref).abs()
    print(f"lm_input full dims: Linf {err.max().item():.3e} L1 {err.mean().item():.3e} (|ref| max {ref.abs().max().item():.2f})")
    assert x.shape == ref.shape == (1 + 1 + P_ + 1 + N, 896)
    assert err.max().item() < 3e-2 and err.mean().item() < 3e-3


def test_orchestrator_with_phoneme_lm_end_to_end():
    """CosyVoice2Model(Qwen2LM_Phoneme_Src2, flow, hift).tts with text = (bpe ids, phoneme factors): the LLM thread, flow and
    HiFT run end to end and yield a finite waveform of 2 * hop samples per generated token."""
    from cosyvoice_amd.config import FlowConfig, HiftConfig
    from cosyvoice_amd.flow import CausalMaskedDiffWithXvec
    from cosyvoice_amd.hift import HiFTGenerator
    from cosyvoice_amd.llm_phoneme import Qwen2LM_Phoneme_Src2
    from cosyvoice_amd.model import CosyVoice2Model
    from cosyvoice_amd.weights import flow_state_dict, hift_state_dict
    lc, pc, fc, hc = LlmConfig.tiny(), PhonemeFrontConfig.tiny(), FlowConfig.tiny(), HiftConfig.tiny()
    llm = Qwen2LM_Phoneme_Src2(lc, pc, dtype=torch.bfloat16, max_batch=2, ctx_max=256, max_out=256)
    llm.load_state_dict(phoneme_lm_state_dict(pc, lc, seed=5, round_to=torch.bfloat16))
    flow = CausalMaskedDiffWithXvec(fc, dtype=torch.float16).load_state_dict(flow_state_dict(fc))
    hift = HiFTGenerator(hc, dtype=torch.float32).load_state_dict(hift_state_dict(hc))
    m = CosyVoice2Model(llm, flow, hift)
    g = torch.Generator().manual_seed(12)
    L, P_, n_p = 4, 9, 6
    pho = torch.stack([torch.randint(0, n, (1, P_), generator=g) for n in (pc.text_token_size, pc.text_tone_size, pc.text_lang_size,
                                                                           pc.text_prsd_size)], dim=-1)
    text = torch.randint(0, lc.vocab_size, (1, L), generator=g)
    e_i, e_p = torch.zeros(1, 0, dtype=torch.int64), torch.zeros(1, 0, 4, dtype=torch.int64)
    ptok = torch.randint(0, lc.speech_token_size, (1, n_p), generator=g, dtype=torch.int32)
    pfeat = torch.clamp(torch.randn(1, 2 * n_p, 80, generator=g) * 2 - 6, -11.5, 2.0)
    out = list(m.tts(text=(text, pho), flow_embedding=torch.randn(1, fc.spk_embed_dim, generator=g),
                     llm_embedding=torch.randn(1, pc.spk_embed_dim, generator=g), prompt_text=(e_i, e_p),
                     llm_prompt_speech_token=ptok, flow_prompt_speech_token=ptok, prompt_speech_feat=pfeat, stream=False))
    assert len(out) == 1
    wav = out[0]["tts_speech"]
    n = wav.shape[1] // (2 * hc.total_upsample)
    assert wav.shape == (1, n * 2 * hc.total_upsample) and 2 * L <= n <= 20 * L and torch.isfinite(wav).all()
