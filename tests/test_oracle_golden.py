"""CPU: the oracle (oracle/*.py) against the golden fixtures minted from the reference itself
(tests/golden/make_golden.py).  fp32 tolerance 1e-5 abs per SURVEY.md §8d."""
import os
import random

import numpy as np
import pytest
import torch

from cosyvoice_amd.config import FlowConfig, HiftConfig, LlmConfig
from cosyvoice_amd.weights import flow_state_dict, hift_state_dict, llm_state_dict
from oracle import flow as of
from oracle import hift as oh
from oracle import llm as ol


def _load(golden_dir, name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, name + ".npz")).items()}


@pytest.mark.parametrize("tag,cfg", [("tiny", HiftConfig.tiny()), ("v2", HiftConfig.v2()), ("v1", HiftConfig.v1())])
def test_hift_decode_f0_source(golden_dir, tag, cfg):
    g = _load(golden_dir, f"hift_{tag}")
    sd = hift_state_dict(cfg)
    wav = oh.decode(sd, cfg, g["mel"], g["s"])
    assert (wav - g["wav"]).abs().max().item() < 1e-5
    assert wav.abs().max().item() < cfg.audio_limit  # fixture is not saturated by the clamp
    f0 = oh.f0_predictor(sd, g["mel"])
    assert (f0 - g["f0"]).abs().max().item() < 1e-3  # values up to ~130 Hz: 1e-5 relative
    f0u = f0[:, None].repeat_interleave(cfg.total_upsample, dim=2).transpose(1, 2)
    src = oh.source_module(sd, cfg, f0u, g["phase_vec"], g["noise"]).transpose(1, 2)
    assert (src - g["src"]).abs().max().item() < 1e-4  # fp32 cumsum ordering (H4)


def test_hift_source_scan_fp64_close_to_fp32():
    cfg = HiftConfig.v2()
    sd = hift_state_dict(cfg)
    f0 = torch.full((1, 1, 20), 220.0)
    f0u = f0.repeat_interleave(cfg.total_upsample, dim=2).transpose(1, 2)
    ph, nz = oh.draw_source_randoms(cfg, 1, f0u.shape[1], seed=3)
    a = oh.source_module(sd, cfg, f0u, ph, nz, torch.float32)
    b = oh.source_module(sd, cfg, f0u, ph, nz, torch.float64)
    assert (a - b).abs().max().item() < 5e-3


def test_flow_estimator_encoder_inference(golden_dir):
    cfg = FlowConfig.tiny()
    sd = flow_state_dict(cfg)
    g = _load(golden_dir, "flow_tiny")
    T = g["est_x"].shape[-1]
    out = of.estimator_forward(sd, cfg, g["est_x"], torch.ones(2, 1, T), g["est_mu"], g["est_t"], g["est_spks"], g["est_cond"])
    assert (out - g["est_out"]).abs().max().item() < 1e-5
    lens = torch.tensor([g["enc_in"].shape[1]])
    e, _ = of.encoder_forward(sd, cfg, g["enc_in"], lens, 0)
    assert (e - g["enc_full"]).abs().max().item() < 1e-5
    e, _ = of.encoder_forward(sd, cfg, g["enc_in"], lens, 4)
    assert (e - g["enc_chunk4"]).abs().max().item() < 1e-5
    assert (of.rand_noise(cfg)[:, :, :64] - g["rand_noise_head"]).abs().max().item() == 0.0
    m = of.inference(sd, cfg, g["token"], g["prompt_token"], g["prompt_feat"], g["embedding"], static_chunk_size=0)
    assert (m - g["mel_full"]).abs().max().item() < 1e-5
    m = of.inference(sd, cfg, g["token"], g["prompt_token"], g["prompt_feat"], g["embedding"], static_chunk_size=4)
    assert (m - g["mel_chunk4"]).abs().max().item() < 1e-5


def test_flow_estimator_full_width_block(golden_dir):
    cfg = FlowConfig(est_n_blocks=1, est_mid_blocks=1, enc_blocks=1, enc_up_blocks=1, vocab_size=64)
    sd = flow_state_dict(cfg)
    g = _load(golden_dir, "flow_est_1block")
    out = of.estimator_forward(sd, cfg, g["est_x"], torch.ones(2, 1, 64), g["est_mu"], g["est_t"], g["est_spks"], g["est_cond"])
    assert (out - g["est_out"]).abs().max().item() < 1e-5


def test_rand_noise_known_head():
    # CausalConditionalCFM.rand_noise starts [-1.12584, -1.15236, -0.25058, ...]
    z = of.rand_noise(FlowConfig())
    assert z.shape == (1, 80, 15000)
    assert torch.allclose(z[0, 0, :3], torch.tensor([-1.12584, -1.15236, -0.25058]), atol=1e-5)


def test_llm_teacher_forced_logp(golden_dir):
    cfg = LlmConfig.tiny()
    sd = llm_state_dict(cfg)
    g = _load(golden_dir, "llm_tiny")
    forced = g["forced"].tolist()
    lp = []
    rnd = random.Random(0)
    toks = list(ol.lm_inference(sd, cfg, g["text"], g["prompt_text"], g["prompt_speech"],
                                uniforms=lambda t: (rnd.random(), rnd.random()), forced_tokens=forced, collect_logp=lp))
    assert toks == forced
    lp = torch.stack(lp)
    assert (lp - g["logps"][: lp.shape[0]]).abs().max().item() < 1e-4


def test_sampler_candidates(golden_dir):
    g = _load(golden_dir, "sampler")
    for r in range(g["scores"].shape[0]):
        _, idx = ol.nucleus_candidates(g["scores"][r])
        c = g["candidates"][r]
        assert idx.tolist() == c[c >= 0].tolist()


def test_ras_repetition_fallback():
    scores = torch.zeros(50)
    scores[7] = 10.0  # nucleus always picks 7
    assert ol.ras_sampling(scores, [1, 2, 3], (0.5, 0.999999)) == 7
    # 7 already in the window -> falls back to sampling the full distribution with the second uniform
    assert ol.ras_sampling(scores, [7, 2, 3], (0.5, 0.0)) == 0


def test_bigvgan_anti_alias_activation(golden_dir):
    from oracle import bigvgan as ob
    from cosyvoice_amd.bigvgan import kaiser_sinc_filter12
    g = _load(golden_dir, "bigvgan_act")
    y = ob.anti_alias_activation(g["x"], g["alpha_log"], g["beta_log"])
    assert (y - g["y"]).abs().max().item() < 1e-5
    assert (ob.kaiser_sinc_filter1d(0.25, 0.3, 12).reshape(-1) - g["up_filter"]).abs().max().item() < 1e-7
    assert (kaiser_sinc_filter12() - g["down_filter"]).abs().max().item() < 1e-7  # the product's own filter table


def test_bigvgan_generator_vs_reference(golden_dir):
    """oracle.bigvgan.bigvgan_forward against the reference's own BigVGAN.forward (golden minted by make_golden.py)."""
    from cosyvoice_amd.config import BigVGANConfig
    from cosyvoice_amd.weights import bigvgan_state_dict
    from oracle import bigvgan as ob
    g = _load(golden_dir, "bigvgan_tiny")
    cfg = BigVGANConfig.tiny()
    sd = bigvgan_state_dict(cfg, seed=int(g["seed"]))
    wav, mel = ob.bigvgan_forward(sd, cfg, g["token"], g["token_len"], g["embedding"])
    assert wav.shape == g["wav"].shape and mel.shape == g["mel"].shape
    assert (wav - g["wav"]).abs().max().item() < 1e-5
    assert (mel - g["mel"]).abs().max().item() < 1e-5


def test_frontend_mel_vs_reference(golden_dir):
    """oracle.frontend.mel_spectrogram against the reference's own function (librosa's mel basis replaced by the restated one
    on both sides: that boundary is unpinned), plus sanity of the restated Slaney basis."""
    from cosyvoice_amd.frontend import align_prompt_24k, slaney_mel_basis
    from oracle import frontend as ofe
    g = _load(golden_dir, "frontend_mel")
    basis = torch.from_numpy(slaney_mel_basis(24000, 1920, 80, 0, 8000))
    mel = ofe.mel_spectrogram(g["y"], basis)
    assert mel.shape == g["mel"].shape
    assert (mel - g["mel"]).abs().max().item() < 1e-4
    assert basis.shape == (80, 961) and (basis >= 0).all()
    peak = basis.argmax(dim=1)
    assert (peak[1:] > peak[:-1]).all()                        # centre frequencies increase
    assert basis[:, 961 * 8000 // 12000 + 2:].abs().max() == 0  # nothing above fmax
    # Slaney normalisation: every triangle has unit area in Hz (bin width 12.5 Hz), up to sampling of the triangle
    area = basis.sum(dim=1) * 12.5
    assert (area - 1.0).abs().max().item() < 0.08
    # cli/frontend.py:148-152
    f, fl, t, tl = align_prompt_24k(torch.zeros(1, 101, 80), torch.zeros(1, 60, dtype=torch.int32))
    assert f.shape[1] == 100 and int(fl) == 100 and t.shape[1] == 50 and int(tl) == 50
    f, fl, t, tl = align_prompt_24k(torch.zeros(1, 100, 80), torch.zeros(1, 37, dtype=torch.int32))
    assert f.shape[1] == 74 and t.shape[1] == 37


def test_phoneme_lm_front_end_vs_reference(golden_dir):
    """oracle.llm_phoneme.phoneme_lm_input against the lm_input the reference's Qwen2LM_Phoneme_Src2.inference hands to its
    Qwen2 stack (golden minted by make_golden.py from the reference module itself)."""
    from cosyvoice_amd.config import LlmConfig, PhonemeFrontConfig
    from cosyvoice_amd.weights import phoneme_lm_state_dict
    from oracle import llm_phoneme as op
    g = _load(golden_dir, "llm_phoneme_tiny")
    lc, pc = LlmConfig.tiny(), PhonemeFrontConfig.tiny()
    sd = phoneme_lm_state_dict(pc, lc, seed=int(g["seed"]))
    x = op.phoneme_lm_input(sd, pc, lc, g["text"], g["pho"], g["prompt_text"], g["prompt_pho"], g["prompt_speech_token"], g["embedding"])
    assert x.shape == g["lm_input"].shape
    assert (x - g["lm_input"]).abs().max().item() < 1e-5


def test_v1_transformer_lm_vs_reference(golden_dir):
    """oracle.llm_v1 (full causal pass) against the log-probabilities of the reference's own TransformerLM.inference, which
    decodes step by step through forward_chunk with an attention cache."""
    from cosyvoice_amd.config import TransformerLMConfig
    from cosyvoice_amd.weights import transformer_lm_state_dict
    from oracle import llm_v1 as o1
    g = _load(golden_dir, "llm_v1_tiny")
    c = TransformerLMConfig.tiny()
    sd = transformer_lm_state_dict(c, seed=int(g["seed"]))
    lp = o1.forced_logp(sd, c, g["text"], g["prompt_text"], g["prompt_speech_token"], g["embedding"], g["forced"].tolist())
    fin = torch.isfinite(g["logp"])
    assert lp.shape == g["logp"].shape and (torch.isinf(lp) == ~fin).all()
    assert (lp[fin] - g["logp"][fin]).abs().max().item() < 1e-4


def test_v1_flow_vs_reference(golden_dir):
    """oracle.flow_v1 against the reference's own MaskedDiffWithXvec (ConformerEncoder, InterpolateRegulator, ConditionalCFM
    with its flow cache, non-causal two-level estimator): a first chunk with an empty cache (head/mid/tail interpolation) and a
    second chunk that inherits the cache (odd mel length: the transposed conv's extra frame is sliced)."""
    from cosyvoice_amd.config import FlowV1Config
    from cosyvoice_amd.weights import flow_v1_state_dict
    from oracle import flow_v1 as o
    g = _load(golden_dir, "flow_v1_tiny")
    c = FlowV1Config.tiny()
    sd = flow_v1_state_dict(c)
    sr = int(g["sample_rate"])
    with torch.no_grad():
        est = o.estimator_forward(sd, c, g["est_x"], g["est_mu"], g["est_t"], g["est_spks"], g["est_cond"])
        assert (est - g["est_out"]).abs().max().item() < 2e-4
        mel1, cache1 = o.inference(sd, c, g["token1"], g["prompt_token"], g["prompt_feat"], g["embedding"],
                                   torch.zeros(1, 80, 0, 2), sr, g["z1"])
        assert mel1.shape == g["mel1"].shape and (mel1 - g["mel1"]).abs().max().item() < 5e-4
        assert (cache1 - g["cache1"]).abs().max().item() < 1e-4
        mel2, cache2 = o.inference(sd, c, g["token2"], g["prompt_token"], g["prompt_feat"], g["embedding"], g["cache1"], sr, g["z2"])
        assert mel2.shape == g["mel2"].shape and (mel2 - g["mel2"]).abs().max().item() < 5e-4
        assert (cache2 - g["cache2"]).abs().max().item() < 1e-4
    print(f"v1 flow oracle vs reference: est {float((est - g['est_out']).abs().max()):.2e}, mel1 {float((mel1 - g['mel1']).abs().max()):.2e}, "
          f"mel2 {float((mel2 - g['mel2']).abs().max()):.2e}")


# ----------------------------------------------------------------------------- round 2 goldens
def test_llm_reference_inference_loop_tiny(golden_dir):
    """The oracle against the fixture minted by driving the reference's OWN Qwen2LM.inference loop (lm_input captured at its
    first forward_one_step, log-prob rows captured at sampling_ids): prefill assembly bit-equal, log-probs 1e-5; and the older
    fixture (re-typed assembly) agrees with it."""
    cfg = LlmConfig.tiny()
    sd = llm_state_dict(cfg)
    g = _load(golden_dir, "llm_tiny_loop")
    lm_in = ol.build_lm_input(sd, cfg, g["text"], g["prompt_text"], g["prompt_speech"])
    assert torch.equal(lm_in[0], g["lm_input"])
    rows = []
    list(ol.lm_inference(sd, cfg, g["text"], g["prompt_text"], g["prompt_speech"], uniforms=lambda t: (0.5, 0.5),
                         forced_tokens=g["forced"].tolist(), collect_logp=rows))
    assert (torch.stack(rows) - g["logps"]).abs().max().item() < 2e-5
    old = _load(golden_dir, "llm_tiny")
    assert torch.equal(old["forced"], g["forced"]) and (old["logps"] - g["logps"]).abs().max().item() < 1e-5


def test_llm_reference_inference_loop_full_size(golden_dir):
    """24 layers, hidden 896, vocab 151 936: 11 teacher-forced log-prob rows of the reference loop vs the oracle."""
    cfg = LlmConfig.full()
    sd = llm_state_dict(cfg)
    g = _load(golden_dir, "llm_full_loop")
    lm_in = ol.build_lm_input(sd, cfg, g["text"], g["prompt_text"], g["prompt_speech"])
    assert torch.equal(lm_in[0, :, :64], g["lm_input"])
    rows = []
    with torch.inference_mode():
        list(ol.lm_inference(sd, cfg, g["text"], g["prompt_text"], g["prompt_speech"], uniforms=lambda t: (0.5, 0.5),
                             forced_tokens=g["forced"].tolist(), collect_logp=rows))
    d = (torch.stack(rows) - g["logps"]).abs().max().item()
    assert d < 5e-5, d


def test_samplers_vs_reference_functions(golden_dir):
    """oracle ras_sampling / non_random_ras_sampling against ids returned by the reference's own functions (utils/common.py:
    105-146) with torch.multinomial replaced by an inverse-CDF draw from recorded uniforms."""
    from oracle.llm_phoneme import non_random_ras_sampling
    g = _load(golden_dir, "sampler_ref")
    n = g["scores"].shape[0]
    for i in range(n):
        u = tuple(g["uniforms"][i].tolist())
        dec = g["decoded"][i].tolist()
        assert ol.ras_sampling(g["scores"][i], dec, u) == int(g["ras"][i]), i
        assert non_random_ras_sampling(g["scores"][i], dec, u, top_k=10, expand_scale=2) == int(g["nrras"][i]), i
        pc, _ = ol.nucleus_candidates(g["scores"][i])
        assert pc.numel() == int(g["ras_n1"][i])
    assert int((g["ras_n2"] > 0).sum()) >= n // 4 and int((g["nrras_n2"] > 0).sum()) >= n // 4   # the fallbacks are exercised


@pytest.mark.parametrize("tag,chunk,key", [("t100", 50, "chunk50"), ("t500", 50, "chunk50"), ("t500", 0, "full")])
def test_flow_full_depth_vs_reference(golden_dir, tag, chunk, key):
    """FULL-depth oracle flow (56 estimator blocks x 10 Euler steps) vs the reference's mel at T = 100 / 500."""
    cfg = FlowConfig.full()
    sd = flow_state_dict(cfg)
    g = _load(golden_dir, "flow_full")
    with torch.inference_mode():
        m = of.inference(sd, cfg, g[f"{tag}_token"], g[f"{tag}_prompt_token"], g[f"{tag}_prompt_feat"], g[f"{tag}_embedding"],
                         static_chunk_size=chunk)
    ref = g[f"{tag}_mel_{key}"]
    d = (m - ref).abs()
    assert d.max().item() < 2e-4 and d.mean().item() < 2e-5, (d.max().item(), d.mean().item())
    assert (m[0].abs().mean(dim=1) - g[f"{tag}_mel_{key}_chan_absmean"]).abs().max().item() < 2e-5


# ----------------------------------------------------------------------------- round 3: goldens at the lengths the C4 bench line runs
def _synth_mel(batch, frames, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.clamp(torch.randn(batch, 80, frames, generator=g) * 2.0 - 6.0, -11.5, 2.0)


@pytest.mark.parametrize("tag,cfg", [("v2", HiftConfig.v2()), ("v1", HiftConfig.v1())])
def test_hift_decode_baseline_length(golden_dir, tag, cfg):
    """oracle decode at 500 (v2) / 861 (v1) frames vs the reference's own HiFTGenerator.decode (every 8th sample exact + block stats)."""
    g = _load(golden_dir, f"hift_{tag}_long")
    frames = int(g["frames"])
    mel = _synth_mel(1, frames, int(g["mel_seed"]))
    s = torch.randn(1, 1, frames * cfg.total_upsample, generator=torch.Generator().manual_seed(int(g["s_seed"]))) * 0.05
    sd = hift_state_dict(cfg)
    wav = oh.decode(sd, cfg, mel, s)
    assert wav.shape[1] == int(g["n_samples"])
    assert (wav[:, ::8] - g["wav_sub8"]).abs().max().item() < 1e-5
    blk = wav[0, : wav.shape[1] // 2400 * 2400].view(-1, 2400).abs()
    assert (blk.mean(dim=1) - g["wav_block_absmean"]).abs().max().item() < 1e-6
    assert (blk.max(dim=1).values - g["wav_block_absmax"]).abs().max().item() < 1e-5
    assert (oh.f0_predictor(sd, mel) - g["f0"]).abs().max().item() < 1e-3


def test_flow_inference_baseline_length(golden_dir):
    """oracle CausalMaskedDiffWithXvec.inference at T = 1000 (250 + 250 tokens, chunk mask 50, 56 blocks x 10 steps) vs the reference."""
    cfg = FlowConfig.full()
    g = _load(golden_dir, "flow_long")
    m = of.inference(flow_state_dict(cfg), cfg, g["token"], g["prompt_token"], g["prompt_feat"], g["embedding"], static_chunk_size=50)
    assert m.shape == (1, 80, 500)
    assert (m[:, :, ::4] - g["mel_sub4"]).abs().max().item() < 2e-4   # fp32 summation order over 10 steps x 56 blocks; values up to 5.9
    assert (m[:, :, ::4] - g["mel_sub4"]).abs().mean().item() < 1e-5
    assert (m[0].abs().mean(dim=1) - g["mel_chan_absmean"]).abs().max().item() < 1e-5
    assert (m[0].mean(dim=0) - g["mel_frame_mean"]).abs().max().item() < 2e-5


def test_llm_long_context_logp(golden_dir):
    """oracle Qwen2 loop at prefill 282 + 250 teacher-forced steps (context 282 -> 532) vs the reference's own inference loop."""
    cfg = LlmConfig.full()
    g = _load(golden_dir, "llm_full_long")
    forced = g["forced"].tolist()
    lp = []
    toks = list(ol.lm_inference(llm_state_dict(cfg), cfg, g["text"], g["prompt_text"], g["prompt_speech"], uniforms=lambda t: (0.5, 0.5),
                                forced_tokens=forced, collect_logp=lp))
    assert toks == forced and len(lp) == 251
    lp = torch.stack(lp)[g["rows"]]
    assert (lp - g["logps"]).abs().max().item() < 1e-4
