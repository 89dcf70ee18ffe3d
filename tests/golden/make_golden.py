#!/usr/bin/env python3
"""Mint the golden fixtures in tests/golden/*.npz by running the REFERENCE's own modules
(/root/reference, read-only) on the build's key-seeded synthetic weights.

Run in the build container only (the reference never travels to the GPU box):
    python tests/golden/make_golden.py

What is imported from the reference: its stage modules, as they are.  What is stubbed:
name-only stand-ins for third-party modules that are absent here and carry NO hot-path
arithmetic (torchmetrics, omegaconf.DictConfig, torchaudio, onnxruntime, conformer) —
and ONE arithmetic-bearing restatement: three diffusers-0.27.2 classes
(Attention / GELU / LoRACompatibleLinear) that flow/components/transformer.py imports.
Estimator goldens therefore pin everything except those three classes ("parity
unpinned" at that boundary).

Fixtures hold inputs + expected outputs only (no reference source).  Weights are not
stored: they are regenerated from cosyvoice_amd.weights (key-seeded).
"""
import importlib.machinery
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install_stubs():
    import transformers  # noqa: F401  (must be imported before torchaudio is stubbed: it probes __spec__)
    from transformers import Qwen2Config, Qwen2ForCausalLM  # noqa: F401
    import torch.nn as nn

    class _Any:  # name-only
        def __init__(self, *a, **k):
            pass

    class DictConfig(dict):
        def __init__(self, content=None, **k):
            super().__init__(content or {})

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

    _stub("torchmetrics")
    _stub("torchmetrics.classification", MulticlassAccuracy=_Any)
    _stub("omegaconf", DictConfig=DictConfig)
    ta = _stub("torchaudio")
    ta.transforms = _stub("torchaudio.transforms")
    ta.compliance = _stub("torchaudio.compliance")
    ta.compliance.kaldi = _stub("torchaudio.compliance.kaldi")
    _stub("onnxruntime", InferenceSession=_Any)
    _stub("conformer", ConformerBlock=nn.Module)

    # ---- arithmetic-bearing restatement of the three diffusers 0.27.2 classes (unpinned) ----
    class GELU(nn.Module):
        def __init__(self, dim_in, dim_out, approximate="none"):
            super().__init__()
            self.proj = nn.Linear(dim_in, dim_out)
            self.approximate = approximate

        def forward(self, x):
            return torch.nn.functional.gelu(self.proj(x), approximate=self.approximate)

    class Attention(nn.Module):
        def __init__(self, query_dim, heads=8, dim_head=64, dropout=0.0, bias=False, cross_attention_dim=None,
                     upcast_attention=False, **kw):
            super().__init__()
            inner = heads * dim_head
            self.heads, self.scale = heads, dim_head ** -0.5
            self.to_q = nn.Linear(query_dim, inner, bias=bias)
            self.to_k = nn.Linear(query_dim, inner, bias=bias)
            self.to_v = nn.Linear(query_dim, inner, bias=bias)
            self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(dropout)])

        def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
            B, T, _ = hidden_states.shape
            h = self.heads
            q = self.to_q(hidden_states).view(B, T, h, -1).transpose(1, 2)
            k = self.to_k(hidden_states).view(B, T, h, -1).transpose(1, 2)
            v = self.to_v(hidden_states).view(B, T, h, -1).transpose(1, 2)
            s = torch.matmul(q, k.transpose(-1, -2)) * self.scale
            if attention_mask is not None:
                s = s + attention_mask.unsqueeze(1)
            o = torch.matmul(s.softmax(-1), v).transpose(1, 2).reshape(B, T, -1)
            return self.to_out[0](o)

    _stub("diffusers")
    _stub("diffusers.models")
    _stub("diffusers.models.attention", GEGLU=_Any, GELU=GELU, AdaLayerNorm=_Any, AdaLayerNormZero=_Any,
          ApproximateGELU=_Any)
    _stub("diffusers.models.attention_processor", Attention=Attention)
    _stub("diffusers.models.lora", LoRACompatibleLinear=nn.Linear)
    _stub("diffusers.models.activations", get_activation=lambda name: {"silu": nn.SiLU(), "swish": nn.SiLU(),
                                                                        "mish": nn.Mish(), "gelu": nn.GELU()}[name])
    _stub("diffusers.utils")
    _stub("diffusers.utils.torch_utils", maybe_allow_in_graph=lambda c: c)
    sys.path.insert(0, REF)


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: tuple(v.shape) for k, v in out.items()})


def summary(t: torch.Tensor):
    t = t.detach().float()
    return np.array([t.mean().item(), t.abs().mean().item(), t.std().item(), t.abs().max().item()], dtype=np.float64)


# ----------------------------------------------------------------------------- HiFT
def build_ref_hift(cfg, sd):
    from cosyvoice.hifigan.f0_predictor import ConvRNNF0Predictor
    from cosyvoice.hifigan.generator import HiFTGenerator
    f0p = ConvRNNF0Predictor(num_class=1, in_channels=cfg.in_channels, cond_channels=cfg.f0_cond_channels)
    m = HiFTGenerator(in_channels=cfg.in_channels, base_channels=cfg.base_channels, nb_harmonics=cfg.nb_harmonics,
                      sampling_rate=cfg.sampling_rate, nsf_alpha=cfg.nsf_alpha, nsf_sigma=cfg.nsf_sigma,
                      nsf_voiced_threshold=cfg.nsf_voiced_threshold, upsample_rates=list(cfg.upsample_rates),
                      upsample_kernel_sizes=list(cfg.upsample_kernel_sizes),
                      istft_params={"n_fft": cfg.n_fft, "hop_len": cfg.hop_len},
                      resblock_kernel_sizes=list(cfg.resblock_kernel_sizes),
                      resblock_dilation_sizes=[list(d) for d in cfg.resblock_dilation_sizes],
                      source_resblock_kernel_sizes=list(cfg.source_resblock_kernel_sizes),
                      source_resblock_dilation_sizes=[list(d) for d in cfg.source_resblock_dilation_sizes],
                      lrelu_slope=cfg.lrelu_slope, audio_limit=cfg.audio_limit, f0_predictor=f0p)
    missing, unexpected = m.load_state_dict(sd, strict=True), None
    return m.eval()


def synth_mel(batch, frames, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.clamp(torch.randn(batch, 80, frames, generator=g) * 2.0 - 6.0, -11.5, 2.0)


def golden_hift():
    from cosyvoice_amd.config import HiftConfig
    from cosyvoice_amd.weights import hift_state_dict
    for tag, cfg, frames in (("tiny", HiftConfig.tiny(), 24), ("v2", HiftConfig.v2(), 20), ("v1", HiftConfig.v1(), 16)):
        sd = hift_state_dict(cfg)
        m = build_ref_hift(cfg, sd)
        mel = synth_mel(1, frames, seed=11)
        g = torch.Generator().manual_seed(12)
        s = torch.randn(1, 1, frames * cfg.total_upsample, generator=g) * 0.05
        with torch.inference_mode():
            wav = m.decode(x=mel, s=s)
            f0 = m.f0_predictor(mel)
            # SineGen + source module with the reference's own RNG: capture its draws by seeding
            torch.manual_seed(1234)
            f0_up = m.f0_upsamp(f0[:, None]).transpose(1, 2)
            src, _, _ = m.m_source(f0_up)
            # replay the reference's draw order to recover the randoms it consumed (generator.py:149-163)
            torch.manual_seed(1234)
            from torch.distributions.uniform import Uniform
            pv = Uniform(low=-np.pi, high=np.pi).sample(sample_shape=(1, cfg.nb_harmonics + 1, 1))
            noise = torch.randn(1, cfg.nb_harmonics + 1, f0_up.shape[1])
        save(f"hift_{tag}", mel=mel, s=s, wav=wav, f0=f0, src=src.transpose(1, 2), phase_vec=pv, noise=noise)


# ----------------------------------------------------------------------------- flow
def build_ref_flow(cfg, sd):
    from omegaconf import DictConfig
    from cosyvoice.flow.decoder import ConditionalDecoder
    from cosyvoice.flow.flow import CausalMaskedDiffWithXvec
    from cosyvoice.flow.flow_matching import CausalConditionalCFM
    from cosyvoice.transformer.upsample_encoder import UpsampleConformerEncoder
    enc = UpsampleConformerEncoder(output_size=cfg.enc_dim, attention_heads=cfg.enc_heads,
                                   linear_units=cfg.enc_linear_units, num_blocks=cfg.enc_blocks, dropout_rate=0.1,
                                   positional_dropout_rate=0.1, attention_dropout_rate=0.1, normalize_before=True,
                                   input_layer="linear", pos_enc_layer_type="rel_pos_espnet",
                                   selfattention_layer_type="rel_selfattn", input_size=cfg.input_size,
                                   use_cnn_module=False, macaron_style=False)
    if cfg.enc_up_blocks != 4:
        enc.up_encoders = torch.nn.ModuleList(list(enc.up_encoders)[:cfg.enc_up_blocks])
    est = ConditionalDecoder(in_channels=cfg.est_in_channels, out_channels=cfg.output_size, causal=True,
                             channels=[cfg.est_channels], dropout=0.0, attention_head_dim=cfg.est_head_dim,
                             n_blocks=cfg.est_n_blocks, num_mid_blocks=cfg.est_mid_blocks, num_heads=cfg.est_heads,
                             act_fn="gelu")
    cfm = CausalConditionalCFM(in_channels=240, n_spks=1, spk_emb_dim=80,
                               cfm_params=DictConfig({"sigma_min": 1e-06, "solver": "euler", "t_scheduler": "cosine",
                                                      "training_cfg_rate": 0.2, "inference_cfg_rate": cfg.inference_cfg_rate,
                                                      "reg_loss_type": "l1"}), estimator=est)
    flow = CausalMaskedDiffWithXvec(input_size=cfg.input_size, output_size=cfg.output_size,
                                    spk_embed_dim=cfg.spk_embed_dim, output_type="mel", vocab_size=cfg.vocab_size,
                                    input_frame_rate=cfg.input_frame_rate, only_mask_loss=True,
                                    token_mel_ratio=cfg.token_mel_ratio, pre_lookahead_len=cfg.pre_lookahead_len,
                                    encoder=enc, decoder=cfm)
    flow.load_state_dict(sd, strict=True)
    return flow.eval()


def golden_flow():
    from cosyvoice_amd.config import FlowConfig
    from cosyvoice_amd.weights import flow_state_dict
    cfg = FlowConfig.tiny()
    sd = flow_state_dict(cfg)
    flow = build_ref_flow(cfg, sd)
    g = torch.Generator().manual_seed(21)
    n_p, n_g = 6, 10
    token = torch.randint(0, cfg.vocab_size, (1, n_g), generator=g, dtype=torch.int32)
    prompt_token = torch.randint(0, cfg.vocab_size, (1, n_p), generator=g, dtype=torch.int32)
    prompt_feat = torch.clamp(torch.randn(1, 2 * n_p, 80, generator=g) * 2 - 6, -11.5, 2.0)
    embedding = torch.randn(1, cfg.spk_embed_dim, generator=g)
    with torch.inference_mode():
        rn = flow.decoder.rand_noise[:, :, :64].clone()
        # (a) estimator alone
        T = 2 * (n_p + n_g)
        x = torch.randn(2, 80, T, generator=g)
        mu = torch.randn(2, 80, T, generator=g)
        cond = torch.randn(2, 80, T, generator=g)
        spks = torch.randn(2, 80, generator=g)
        t = torch.tensor([0.3, 0.3])
        mask = torch.ones(2, 1, T)
        est_out = flow.decoder.estimator(x, mask, mu, t, spks, cond)
        # (b) encoder alone, full attention and chunk-16 attention
        xs = torch.randn(1, n_p + n_g, cfg.input_size, generator=g)
        lens = torch.tensor([n_p + n_g])
        flow.encoder.static_chunk_size = 0
        enc_full, _ = flow.encoder(xs, lens)
        flow.encoder.static_chunk_size = 4
        enc_chunk, _ = flow.encoder(xs, lens)
        # (c) whole inference, CosyVoiceModel wiring (static_chunk_size 0) and CosyVoice2Model wiring (chunk 4 here)
        flow.encoder.static_chunk_size = 0
        mel_full, _ = flow.inference(token=token, token_len=torch.tensor([n_g]), prompt_token=prompt_token,
                                     prompt_token_len=torch.tensor([n_p]), prompt_feat=prompt_feat,
                                     prompt_feat_len=torch.tensor([2 * n_p]), embedding=embedding)
        flow.encoder.static_chunk_size = 4
        mel_chunk, _ = flow.inference(token=token, token_len=torch.tensor([n_g]), prompt_token=prompt_token,
                                      prompt_token_len=torch.tensor([n_p]), prompt_feat=prompt_feat,
                                      prompt_feat_len=torch.tensor([2 * n_p]), embedding=embedding)
    save("flow_tiny", token=token, prompt_token=prompt_token, prompt_feat=prompt_feat, embedding=embedding,
         rand_noise_head=rn, est_x=x, est_mu=mu, est_cond=cond, est_spks=spks, est_t=t, est_out=est_out,
         enc_in=xs, enc_full=enc_full, enc_chunk4=enc_chunk, mel_full=mel_full, mel_chunk4=mel_chunk)

    # full-shape single transformer block + resnet block of the estimator (T=64) for layer-level pinning
    cfgf = FlowConfig(est_n_blocks=1, est_mid_blocks=1, enc_blocks=1, enc_up_blocks=1, vocab_size=64)
    sdf = flow_state_dict(cfgf)
    flowf = build_ref_flow(cfgf, sdf)
    with torch.inference_mode():
        T = 64
        x = torch.randn(2, 80, T, generator=g); mu = torch.randn(2, 80, T, generator=g)
        cond = torch.randn(2, 80, T, generator=g); spks = torch.randn(2, 80, generator=g)
        t = torch.tensor([0.7, 0.7]); mask = torch.ones(2, 1, T)
        out = flowf.decoder.estimator(x, mask, mu, t, spks, cond)
    save("flow_est_1block", est_x=x, est_mu=mu, est_cond=cond, est_spks=spks, est_t=t, est_out=out)


# ----------------------------------------------------------------------------- llm
def build_ref_llm(cfg, sd):
    from cosyvoice.llm.llm import Qwen2Encoder, Qwen2LM
    from cosyvoice.utils.common import ras_sampling
    d = tempfile.mkdtemp()
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(cfg.hf_config_dict(), f)
    enc = Qwen2Encoder(d)
    lm = Qwen2LM(llm_input_size=cfg.hidden_size, llm_output_size=cfg.hidden_size, speech_token_size=cfg.speech_token_size,
                 llm=enc, sampling=ras_sampling)
    missing = lm.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys, missing.unexpected_keys
    assert all("rotary" in k or "inv_freq" in k for k in missing.missing_keys), missing.missing_keys
    return lm.eval()


def golden_llm():
    from cosyvoice_amd.config import LlmConfig
    from cosyvoice_amd.weights import llm_state_dict
    cfg = LlmConfig.tiny()
    sd = llm_state_dict(cfg)
    lm = build_ref_llm(cfg, sd)
    g = torch.Generator().manual_seed(31)
    text = torch.randint(0, cfg.vocab_size, (1, 6), generator=g, dtype=torch.int32)
    prompt_text = torch.randint(0, cfg.vocab_size, (1, 4), generator=g, dtype=torch.int32)
    prompt_speech = torch.randint(0, cfg.speech_token_size, (1, 9), generator=g, dtype=torch.int32)
    forced = torch.randint(0, cfg.speech_token_size, (12,), generator=g).tolist()
    # teacher-forced log-probs through the reference's own forward_one_step / llm_decoder (llm.py:861-874)
    with torch.inference_mode():
        t = torch.cat([prompt_text, text], dim=1)
        te = lm.llm.model.model.embed_tokens(t)
        sos = lm.llm_embedding.weight[lm.sos_eos].reshape(1, 1, -1)
        task = lm.llm_embedding.weight[lm.task_id].reshape(1, 1, -1)
        pe = lm.speech_embedding(prompt_speech)
        lm_input = torch.cat([sos, te, task, pe], dim=1)
        cache = None
        logps = []
        past = 0
        for i in range(len(forced) + 1):
            # NOTE: the reference's forward_one_step passes attention_mask = masks[:, -1, :], i.e. a mask of
            # the CURRENT chunk length only (llm.py:755,862-864).  transformers<=4.5x ignores an all-ones mask
            # (full causal attention over the cache — also what the reference's own graph path does,
            # qwen2_5.py:154-162: no mask at decode); the transformers 5.15 installed here mis-reads the short
            # mask at decode steps.  The golden therefore calls the same HF model with a full-length mask.
            L = lm_input.shape[1]
            outs = lm.llm.model(inputs_embeds=lm_input, attention_mask=torch.ones(1, past + L, dtype=torch.bool),
                                output_hidden_states=True, return_dict=True, use_cache=True, past_key_values=cache)
            y, cache = outs.hidden_states[-1], outs.past_key_values
            past += L
            logps.append(lm.llm_decoder(y[:, -1]).log_softmax(dim=-1))
            if i < len(forced):
                lm_input = lm.speech_embedding.weight[forced[i]].reshape(1, 1, -1)
        logps = torch.cat(logps, 0)
    save("llm_tiny", text=text, prompt_text=prompt_text, prompt_speech=prompt_speech, forced=np.array(forced),
         logps=logps)

    # sampler candidate sets (deterministic part of nucleus_sampling, utils/common.py:126-137)
    from cosyvoice.utils import common as C
    scores = torch.randn(4, 300, generator=g) * 2.0
    cand = []
    for r in range(4):
        sv, si = scores[r].softmax(dim=0).sort(descending=True, stable=True)
        cum, n = 0.0, 0
        while n < 25 and cum < 0.8:
            cum += sv[n].item(); n += 1
        row = np.full(25, -1, dtype=np.int64); row[:n] = si[:n].numpy()
        cand.append(row)
    save("sampler", scores=scores, candidates=np.stack(cand))


# ----------------------------------------------------------------------------- BigVGAN activation
def golden_bigvgan_act():
    """Reference torch path of the fused kernel: alias_free_activation/torch/act.py Activation1d + nnet SnakeBeta."""
    from cosyvoice.BigVGAN.alias_free_activation.torch.act import Activation1d
    from cosyvoice.BigVGAN.nnet.activations import SnakeBeta
    g = torch.Generator().manual_seed(41)
    C, T = 6, 301
    act = SnakeBeta(C, alpha_logscale=True)
    with torch.no_grad():
        act.alpha.copy_(torch.randn(C, generator=g) * 0.5)
        act.beta.copy_(torch.randn(C, generator=g) * 0.5)
    m = Activation1d(activation=act)
    x = torch.randn(2, C, T, generator=g) * 1.5
    with torch.inference_mode():
        y = m(x)
    save("bigvgan_act", x=x, alpha_log=act.alpha.detach(), beta_log=act.beta.detach(), y=y,
         up_filter=m.upsample.filter.reshape(-1), down_filter=m.downsample.lowpass.filter.reshape(-1))


def golden_bigvgan_model():
    """The reference's BigVGAN generator itself (BigVGAN/bigvgan.py), encoder1 = encoder2 = None, torch activation path.
    Its module imports the CUDA-extension loader at import time (cuda/activation1d.py:10 shells out to nvcc): that one
    module is replaced by a name-only stub — use_cuda_kernel=False never touches it."""
    from cosyvoice_amd.config import BigVGANConfig
    from cosyvoice_amd.weights import bigvgan_state_dict
    class _NameOnly:
        pass
    _stub("cosyvoice.BigVGAN.alias_free_activation.cuda")
    _stub("cosyvoice.BigVGAN.alias_free_activation.cuda.activation1d", Activation1d=_NameOnly)
    from cosyvoice.BigVGAN.bigvgan import BigVGAN
    cfg = BigVGANConfig.tiny()
    sd = bigvgan_state_dict(cfg, seed=5)
    m = BigVGAN(vocab_size=cfg.vocab_size, input_size=cfg.input_size, output_size=cfg.output_size, mel_bin=cfg.mel_bin,
                upsample_rates=list(cfg.upsample_rates), upsample_kernel_sizes=list(cfg.upsample_kernel_sizes),
                upsample_initial_channel=cfg.upsample_initial_channel, resblock_kernel_sizes=list(cfg.resblock_kernel_sizes),
                resblock_dilation_sizes=[list(d) for d in cfg.resblock_dilation_sizes],
                speaker_embedding_dim=cfg.speaker_embedding_dim, use_cuda_kernel=False)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    missing = [k for k in missing if not k.endswith("filter")]   # the kaiser-sinc filters are buffers rebuilt at construction
    assert not missing and not unexpected, (missing, unexpected)
    m.eval()
    g = torch.Generator().manual_seed(77)
    B, N = 2, 21
    token = torch.randint(0, cfg.vocab_size, (B, N), generator=g)
    token_len = torch.tensor([N, N - 6])
    emb = torch.randn(B, cfg.speaker_embedding_dim, generator=g)
    with torch.inference_mode():
        wav, (mel, _) = m(dict(speech_token=token, speech_token_len=token_len, embedding=emb), torch.device("cpu"))
    save("bigvgan_tiny", token=token.to(torch.int32), token_len=token_len.to(torch.int32), embedding=emb, wav=wav, mel=mel, seed=np.array(5))


def golden_frontend_mel():
    """The reference's own mel_spectrogram (dataset/processor_kaldidata.py:37-74) at the CosyVoice2 configuration, with
    librosa.filters.mel (absent, unpinned) replaced by the build's restatement of its default algorithm: pins everything
    after the basis (reflect pad, hann STFT, magnitude, projection, log-compression)."""
    from cosyvoice_amd.frontend import slaney_mel_basis
    _stub("librosa")
    _stub("librosa.filters", mel=lambda sr, n_fft, n_mels, fmin, fmax: slaney_mel_basis(sr, n_fft, n_mels, fmin, fmax))
    sys.modules["torchaudio"].set_audio_backend = lambda *a, **k: None
    from cosyvoice.dataset.processor_kaldidata import mel_spectrogram
    g = torch.Generator().manual_seed(123)
    t = torch.arange(2 * 24000 + 317) / 24000.0
    y = 0.4 * torch.sin(2 * torch.pi * 220.0 * t) + 0.2 * torch.sin(2 * torch.pi * 3100.0 * t * (1 + 0.1 * t)) \
        + 0.05 * torch.randn(t.shape, generator=g)
    y = torch.stack([y, torch.flip(y, dims=[0]) * 0.5]).clamp(-1, 1)
    with torch.inference_mode():
        mel = mel_spectrogram(y, n_fft=1920, num_mels=80, sampling_rate=24000, hop_size=480, win_size=1920, fmin=0, fmax=8000,
                              center=False)
    save("frontend_mel", y=y, mel=mel)


def golden_llm_phoneme():
    """Qwen2LM_Phoneme_Src2 (llm/llm.py:1450-1772), the LM every recipe of the fork uses: the reference module itself
    (reference ConformerEncoder text encoder, reference DecoderLayer, HF Qwen2 from a config-only directory) run up to the
    first forward_one_step call; the prefill embedding sequence lm_input it hands to the Qwen2 stack is the golden."""
    from cosyvoice_amd.config import LlmConfig, PhonemeFrontConfig
    from cosyvoice_amd.weights import phoneme_lm_state_dict
    from cosyvoice.llm.llm import Qwen2Encoder, Qwen2LM_Phoneme_Src2
    from cosyvoice.transformer.encoder import ConformerEncoder
    from cosyvoice.utils.common import non_random_ras_sampling
    lc, pc = LlmConfig.tiny(), PhonemeFrontConfig.tiny()
    sd = phoneme_lm_state_dict(pc, lc, seed=31)
    with tempfile.TemporaryDirectory() as d:
        json.dump(lc.hf_config_dict(), open(os.path.join(d, "config.json"), "w"))
        enc = ConformerEncoder(input_size=pc.input_size, output_size=pc.enc_dim, attention_heads=pc.enc_heads,
                               linear_units=pc.enc_linear_units, num_blocks=pc.enc_blocks, dropout_rate=0.1,
                               positional_dropout_rate=0.1, attention_dropout_rate=0.0, normalize_before=True, input_layer="linear",
                               pos_enc_layer_type="rel_pos_espnet", selfattention_layer_type="rel_selfattn", use_cnn_module=False,
                               macaron_style=False, use_dynamic_chunk=False, use_dynamic_left_chunk=False, static_chunk_size=-1)
        m = Qwen2LM_Phoneme_Src2(text_encoder_input_size=pc.input_size, llm_input_size=lc.hidden_size, llm_output_size=lc.hidden_size,
                                 text_token_size=pc.text_token_size, text_token_dim=pc.text_token_dim, text_tone_size=pc.text_tone_size,
                                 text_tone_dim=pc.text_tone_dim, text_lang_size=pc.text_lang_size, text_lang_dim=pc.text_lang_dim,
                                 text_prsd_size=pc.text_prsd_size, text_prsd_dim=pc.text_prsd_dim,
                                 speech_token_size=lc.speech_token_size, text_encoder=enc, llm=Qwen2Encoder(d),
                                 sampling=non_random_ras_sampling, spk_embed_dim=pc.spk_embed_dim)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("rotary" in k or "inv_freq" in k for k in missing), (missing, unexpected)
    m.eval()
    g = torch.Generator().manual_seed(17)
    L, P, Lp, Pp, N = 6, 11, 3, 5, 7

    def pho_ids(n):
        return torch.stack([torch.randint(0, pc.text_token_size, (1, n), generator=g), torch.randint(0, pc.text_tone_size, (1, n), generator=g),
                            torch.randint(0, pc.text_lang_size, (1, n), generator=g), torch.randint(0, pc.text_prsd_size, (1, n), generator=g)], dim=-1)

    text = torch.randint(0, lc.vocab_size, (1, L), generator=g)
    ptext = torch.randint(0, lc.vocab_size, (1, Lp), generator=g)
    pho, ppho = pho_ids(P), pho_ids(Pp)
    pspeech = torch.randint(0, lc.speech_token_size, (1, N), generator=g)
    emb = torch.randn(1, pc.spk_embed_dim, generator=g)
    captured = {}
    orig = m.llm.forward_one_step

    class _Stop(Exception):
        pass

    def spy(xs, masks, cache=None):
        captured["lm_input"] = xs.detach().clone()
        raise _Stop()

    m.llm.forward_one_step = spy
    try:
        with torch.inference_mode():
            next(m.inference(text=(text, pho), text_len=(torch.tensor([L]), torch.tensor([P])), prompt_text=(ptext, ppho),
                             prompt_text_len=(torch.tensor([Lp]), torch.tensor([Pp])), prompt_speech_token=pspeech,
                             prompt_speech_token_len=torch.tensor([N]), embedding=emb))
    except (_Stop, RuntimeError) as e:   # a generator converts the exception of its body into RuntimeError
        if "lm_input" not in captured:
            raise e
    m.llm.forward_one_step = orig
    save("llm_phoneme_tiny", text=text.to(torch.int32), pho=pho.to(torch.int32), prompt_text=ptext.to(torch.int32),
         prompt_pho=ppho.to(torch.int32), prompt_speech_token=pspeech.to(torch.int32), embedding=emb,
         lm_input=captured["lm_input"], seed=np.array(31))


def golden_v1_orchestrator():
    """Chunk schedule of the reference's CosyVoiceModel (v1 wiring, cli/model.py:27-292) driving CosyVoice2 modules, as the
    fork does: vc() over 210 source tokens (a 60-token tail: the reference cross-fade needs the last chunk to hold at least mel_overlap_len = 68 frames), streaming and not — the number of samples of every yielded chunk (the waveforms
    themselves carry the vocoder's random source phases and are not comparable)."""
    from cosyvoice.cli.model import CosyVoiceModel
    from cosyvoice_amd.config import FlowConfig, HiftConfig
    from cosyvoice_amd.weights import flow_state_dict, hift_state_dict
    fc, hc = FlowConfig.tiny(), HiftConfig.v1()
    flow = build_ref_flow(fc, flow_state_dict(fc))
    hift = build_ref_hift(hc, hift_state_dict(hc))

    class _NoLLM(torch.nn.Module):
        fp16 = False

    m = CosyVoiceModel(_NoLLM(), flow, hift, fp16=False, sr=22050)
    m.device = torch.device("cpu")
    g = torch.Generator().manual_seed(99)
    n_p = 12
    src = torch.randint(0, fc.vocab_size, (1, 210), generator=g, dtype=torch.int32)
    ptok = torch.randint(0, fc.vocab_size, (1, n_p), generator=g, dtype=torch.int32)
    pfeat = torch.clamp(torch.randn(1, 2 * n_p, 80, generator=g) * 2 - 6, -11.5, 2.0)
    emb = torch.randn(1, fc.spk_embed_dim, generator=g)
    out = {}
    for stream in (True, False):
        torch.manual_seed(0)
        with torch.inference_mode():
            chunks = [o["tts_speech"] for o in m.vc(src, ptok, pfeat, emb, stream=stream)]
        out["stream" if stream else "full"] = np.array([c.shape[1] for c in chunks], dtype=np.int64)
        assert all(torch.isfinite(c).all() for c in chunks)
    save("v1_orchestrator", source_speech_token=src, prompt_token=ptok, prompt_feat=pfeat, embedding=emb,
         stream_chunk_samples=out["stream"], full_samples=out["full"])

    # the genuine v1 stack: reference MaskedDiffWithXvec (50 Hz tokens, flow cache) + HiFT v1 under the same orchestrator
    from cosyvoice_amd.config import FlowV1Config
    from cosyvoice_amd.weights import flow_v1_state_dict
    f1 = FlowV1Config.tiny()
    m = CosyVoiceModel(_NoLLM(), build_ref_flow_v1(f1, flow_v1_state_dict(f1)), hift, fp16=False, sr=22050)
    m.device = torch.device("cpu")
    src = torch.randint(0, f1.vocab_size, (1, 330), generator=g, dtype=torch.int32)
    ptok = torch.randint(0, f1.vocab_size, (1, 2 * n_p), generator=g, dtype=torch.int32)
    pfeat = torch.clamp(torch.randn(1, 20, 80, generator=g) * 2 - 6, -11.5, 2.0)
    out = {}
    for stream in (True, False):
        torch.manual_seed(0)
        with torch.inference_mode():
            chunks = [o["tts_speech"] for o in m.vc(src, ptok, pfeat, emb, stream=stream)]
        out["stream" if stream else "full"] = np.array([c.shape[1] for c in chunks], dtype=np.int64)
        assert all(torch.isfinite(c).all() for c in chunks)
    save("v1_orchestrator_v1flow", source_speech_token=src, prompt_token=ptok, prompt_feat=pfeat, embedding=emb,
         stream_chunk_samples=out["stream"], full_samples=out["full"], token_min_hop_len=np.array(m.token_min_hop_len),
         mel_overlap_len=np.array(m.mel_overlap_len))


def golden_llm_v1():
    """CosyVoice-v1 TransformerLM (llm/llm.py:41-237): the reference module itself (reference ConformerEncoder text encoder,
    reference TransformerEncoder with its forward_chunk attention cache) on key-seeded weights; sampling_ids is replaced by a
    recorder that stores the log-probabilities of every step and returns teacher-forced ids."""
    from cosyvoice_amd.config import TransformerLMConfig
    from cosyvoice_amd.weights import transformer_lm_state_dict
    from cosyvoice.llm.llm import TransformerLM
    from cosyvoice.transformer.encoder import ConformerEncoder, TransformerEncoder
    from cosyvoice.utils.common import non_random_ras_sampling
    c = TransformerLMConfig.tiny()
    sd = transformer_lm_state_dict(c, seed=41)
    enc = ConformerEncoder(input_size=c.text_encoder_input_size, output_size=c.enc_dim, attention_heads=c.enc_heads,
                           linear_units=c.enc_linear_units, num_blocks=c.enc_blocks, dropout_rate=0.1, positional_dropout_rate=0.1,
                           attention_dropout_rate=0.0, normalize_before=True, input_layer="linear", pos_enc_layer_type="rel_pos_espnet",
                           selfattention_layer_type="rel_selfattn", use_cnn_module=False, macaron_style=False, use_dynamic_chunk=False,
                           use_dynamic_left_chunk=False, static_chunk_size=1)
    llm = TransformerEncoder(input_size=c.llm_dim, output_size=c.llm_dim, attention_heads=c.llm_heads, linear_units=c.llm_linear_units,
                             num_blocks=c.llm_blocks, dropout_rate=0.1, positional_dropout_rate=0.1, attention_dropout_rate=0.0,
                             input_layer="linear_legacy", pos_enc_layer_type="rel_pos_espnet", selfattention_layer_type="rel_selfattn",
                             static_chunk_size=1)
    m = TransformerLM(text_encoder_input_size=c.text_encoder_input_size, llm_input_size=c.llm_dim, llm_output_size=c.llm_dim,
                      text_token_size=c.text_token_size, speech_token_size=c.speech_token_size, text_encoder=enc, llm=llm,
                      sampling=non_random_ras_sampling, spk_embed_dim=c.spk_embed_dim)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and not [k for k in missing if "pe" not in k.split(".")[-1]], (missing, unexpected)
    m.eval()
    m.fp16 = False
    g = torch.Generator().manual_seed(29)
    L, Lp, N = 7, 3, 9
    text = torch.randint(0, c.text_token_size, (1, L), generator=g)
    ptext = torch.randint(0, c.text_token_size, (1, Lp), generator=g)
    pspeech = torch.randint(0, c.speech_token_size, (1, N), generator=g)
    emb = torch.randn(1, c.spk_embed_dim, generator=g)
    forced = torch.randint(0, c.speech_token_size, (12,), generator=g).tolist()
    rows = []

    def recorder(weighted_scores, decoded_tokens, sampling, ignore_eos=True):
        rows.append(weighted_scores.detach().clone())
        i = len(decoded_tokens)
        return torch.tensor(forced[i] if i < len(forced) else c.speech_token_size)

    m.sampling_ids = recorder
    with torch.inference_mode():
        toks = list(m.inference(text=text, text_len=torch.tensor([L]), prompt_text=ptext, prompt_text_len=torch.tensor([Lp]),
                                prompt_speech_token=pspeech, prompt_speech_token_len=torch.tensor([N]), embedding=emb))
    assert toks == forced and len(rows) == len(forced) + 1
    save("llm_v1_tiny", text=text.to(torch.int32), prompt_text=ptext.to(torch.int32), prompt_speech_token=pspeech.to(torch.int32),
         embedding=emb, forced=np.array(forced, dtype=np.int32), logp=torch.stack(rows), seed=np.array(41))


def build_ref_flow_v1(cfg, sd):
    from omegaconf import DictConfig
    from cosyvoice.flow.decoder import ConditionalDecoder
    from cosyvoice.flow.flow import MaskedDiffWithXvec
    from cosyvoice.flow.flow_matching import ConditionalCFM
    from cosyvoice.flow.length_regulator import InterpolateRegulator
    from cosyvoice.transformer.encoder import ConformerEncoder
    enc = ConformerEncoder(output_size=cfg.enc_dim, attention_heads=cfg.enc_heads, linear_units=cfg.enc_linear_units,
                           num_blocks=cfg.enc_blocks, dropout_rate=0.1, positional_dropout_rate=0.1, attention_dropout_rate=0.1,
                           normalize_before=True, input_layer="linear", pos_enc_layer_type="rel_pos_espnet",
                           selfattention_layer_type="rel_selfattn", input_size=cfg.input_size, use_cnn_module=False,
                           macaron_style=False)
    est = ConditionalDecoder(in_channels=cfg.est_in_channels, out_channels=cfg.output_size, channels=[cfg.est_channels] * 2,
                             dropout=0.0, attention_head_dim=cfg.est_head_dim, n_blocks=cfg.est_n_blocks,
                             num_mid_blocks=cfg.est_mid_blocks, num_heads=cfg.est_heads, act_fn="gelu")
    cfm = ConditionalCFM(in_channels=240, n_spks=1, spk_emb_dim=80,
                         cfm_params=DictConfig({"sigma_min": 1e-06, "solver": "euler", "t_scheduler": "cosine",
                                                "training_cfg_rate": 0.2, "inference_cfg_rate": cfg.inference_cfg_rate,
                                                "reg_loss_type": "l1"}), estimator=est)
    flow = MaskedDiffWithXvec(input_size=cfg.input_size, output_size=cfg.output_size, spk_embed_dim=cfg.spk_embed_dim,
                              output_type="mel", vocab_size=cfg.vocab_size, input_frame_rate=cfg.input_frame_rate,
                              only_mask_loss=True, encoder=enc,
                              length_regulator=InterpolateRegulator(channels=cfg.output_size, sampling_ratios=[1] * cfg.reg_layers),
                              decoder=cfm)
    flow.load_state_dict(sd, strict=True)
    return flow.eval()


def golden_flow_v1():
    """CosyVoice-v1 MaskedDiffWithXvec (flow/flow.py:25-160): the reference modules themselves — ConformerEncoder,
    InterpolateRegulator, ConditionalCFM with its flow cache, non-causal ConditionalDecoder channels=[C, C] (through the same
    restated diffusers classes as golden_flow) — on key-seeded weights.  The torch.randn_like draw of flow_matching.py:56 is
    reproduced by re-seeding the global generator (checked against the z half of the returned flow cache)."""
    from cosyvoice_amd.config import FlowV1Config
    from cosyvoice_amd.weights import flow_v1_state_dict
    cfg = FlowV1Config.tiny()
    flow = build_ref_flow_v1(cfg, flow_v1_state_dict(cfg))
    est = flow.decoder.estimator
    g = torch.Generator().manual_seed(33)
    sr = 22050
    n_p, n_g1, n_g2, t1 = 12, 50, 30, 20
    prompt_token = torch.randint(0, cfg.vocab_size, (1, n_p), generator=g, dtype=torch.int32)
    token1 = torch.randint(0, cfg.vocab_size, (1, n_g1), generator=g, dtype=torch.int32)
    token2 = torch.randint(0, cfg.vocab_size, (1, n_g2), generator=g, dtype=torch.int32)
    prompt_feat = torch.clamp(torch.randn(1, t1, 80, generator=g) * 2 - 6, -11.5, 2.0)
    embedding = torch.randn(1, cfg.spk_embed_dim, generator=g)

    def call(token, cache, seed):
        torch.manual_seed(seed)
        mel, new_cache = flow.inference(token=token, token_len=torch.tensor([token.shape[1]]), prompt_token=prompt_token,
                                        prompt_token_len=torch.tensor([n_p]), prompt_feat=prompt_feat,
                                        prompt_feat_len=torch.tensor([t1]), embedding=embedding, flow_cache=cache, sample_rate=sr)
        torch.manual_seed(seed)
        z = torch.randn(1, 80, t1 + mel.shape[2])
        return mel, new_cache, z

    with torch.inference_mode():
        mel1, cache1, z1 = call(token1, torch.zeros(1, 80, 0, 2), 101)
        assert torch.equal(cache1[:, :, :t1, 0], z1[:, :, :t1]) and torch.equal(cache1[:, :, t1:, 0], z1[:, :, -34:])
        mel2, cache2, z2 = call(token2, cache1, 102)          # T = 71 (odd): the up path slices the transposed conv's extra frame
        T = 37
        x = torch.randn(2, 80, T, generator=g); mu = torch.randn(2, 80, T, generator=g)
        cond = torch.randn(2, 80, T, generator=g); spks = torch.randn(2, 80, generator=g)
        t = torch.tensor([0.45, 0.45])
        est_out = est(x, torch.ones(2, 1, T), mu, t, spks, cond)
    save("flow_v1_tiny", prompt_token=prompt_token, token1=token1, token2=token2, prompt_feat=prompt_feat, embedding=embedding,
         sample_rate=np.array(sr), z1=z1, mel1=mel1, cache1=cache1, z2=z2, mel2=mel2, cache2=cache2,
         est_x=x, est_mu=mu, est_cond=cond, est_spks=spks, est_t=t, est_out=est_out)


# ----------------------------------------------------------------------------- round 2: full-depth / reference-loop goldens
def _ref_lm_inference_capture(lm, text, prompt_text, prompt_speech, forced, eos):
    """Drive the REFERENCE's own Qwen2LM.inference (llm/llm.py:823-874) teacher-forced: `forward_one_step` is wrapped to (a)
    record the prefill sequence lm_input it receives on its first call and (b) hand HF a full-length attention mask (the
    reference passes a current-chunk-length mask, llm.py:755,862; transformers 5.15 mis-reads that short mask at decode steps,
    older releases ignore an all-ones mask = full causal attention over the cache, which is also what the reference's own
    graph path does, qwen2_5.py:154-162); `sampling_ids` is replaced by a recorder that stores every log-prob row and returns
    the forced id.  Everything else — the input assembly, llm_decoder, log_softmax, the loop bookkeeping — is the reference's."""
    captured, rows = {}, []
    enc = lm.llm

    def fos(xs, masks, cache=None):
        if "lm_input" not in captured:
            captured["lm_input"] = xs.detach().clone()
        past = 0 if cache is None else cache.get_seq_length()
        outs = enc.model(inputs_embeds=xs, attention_mask=torch.ones(1, past + xs.shape[1], dtype=torch.bool),
                         output_hidden_states=True, return_dict=True, use_cache=True, past_key_values=cache)
        return outs.hidden_states[-1], outs.past_key_values

    def recorder(weighted_scores, decoded_tokens, sampling, ignore_eos=True):
        rows.append(weighted_scores.detach().clone())
        i = len(decoded_tokens)
        return torch.tensor([forced[i] if i < len(forced) else eos])

    enc.forward_one_step, lm.sampling_ids = fos, recorder
    with torch.inference_mode():
        toks = list(lm.inference(text=text, text_len=torch.tensor([text.shape[1]]), prompt_text=prompt_text,
                                 prompt_text_len=torch.tensor([prompt_text.shape[1]]), prompt_speech_token=prompt_speech,
                                 prompt_speech_token_len=torch.tensor([prompt_speech.shape[1]]), embedding=torch.zeros(0, 192)))
    assert toks == list(forced) and len(rows) == len(forced) + 1, (toks, len(rows))
    return captured["lm_input"], torch.stack(rows)


def golden_llm_loop():
    """(a) tiny config: lm_input + teacher-forced log-probs through the reference's own inference() loop (supersedes the re-typed
    assembly of golden_llm: the two must agree); (b) FULL size (24 layers, hidden 896, vocab 151 936): the same for 10 steps."""
    from cosyvoice_amd.config import LlmConfig
    from cosyvoice_amd.weights import llm_state_dict
    for tag, cfg, n_text, n_pt, n_ps, n_forced in (("tiny", LlmConfig.tiny(), 6, 4, 9, 12), ("full", LlmConfig.full(), 12, 6, 30, 10)):
        lm = build_ref_llm(cfg, llm_state_dict(cfg))
        g = torch.Generator().manual_seed(31 if tag == "tiny" else 47)
        text = torch.randint(0, cfg.vocab_size, (1, n_text), generator=g, dtype=torch.int32)
        prompt_text = torch.randint(0, cfg.vocab_size, (1, n_pt), generator=g, dtype=torch.int32)
        prompt_speech = torch.randint(0, cfg.speech_token_size, (1, n_ps), generator=g, dtype=torch.int32)
        forced = torch.randint(0, cfg.speech_token_size, (n_forced,), generator=g).tolist()
        lm_input, logps = _ref_lm_inference_capture(lm, text, prompt_text, prompt_speech, forced, cfg.speech_token_size)
        save(f"llm_{tag}_loop", text=text, prompt_text=prompt_text, prompt_speech=prompt_speech, forced=np.array(forced),
             lm_input=lm_input[0] if tag == "tiny" else lm_input[0, :, :64], lm_input_summary=summary(lm_input), logps=logps)
        del lm


LLM_LONG_ROWS = (0, 1, 63, 127, 229, 230, 231, 249, 250)


def golden_llm_long():
    """FULL-size LM at the context the C4 bench line runs: prefill L = 1 + 10 + 20 + 1 + 250 = 282, then 250
    teacher-forced steps through the reference's own Qwen2LM.inference loop (llm/llm.py:823-874), so the attended context grows
    282 -> 532 and crosses 512 keys at row 230.  Kept: the log-prob rows LLM_LONG_ROWS (row i = distribution after i forced tokens,
    i.e. over 282 + i keys) and the last-position summary of lm_input."""
    from cosyvoice_amd.config import LlmConfig
    from cosyvoice_amd.weights import llm_state_dict
    cfg = LlmConfig.full()
    lm = build_ref_llm(cfg, llm_state_dict(cfg))
    g = torch.Generator().manual_seed(53)
    text = torch.randint(0, cfg.vocab_size, (1, 20), generator=g, dtype=torch.int32)
    prompt_text = torch.randint(0, cfg.vocab_size, (1, 10), generator=g, dtype=torch.int32)
    prompt_speech = torch.randint(0, cfg.speech_token_size, (1, 250), generator=g, dtype=torch.int32)
    forced = torch.randint(0, cfg.speech_token_size, (250,), generator=g).tolist()
    lm_input, logps = _ref_lm_inference_capture(lm, text, prompt_text, prompt_speech, forced, cfg.speech_token_size)
    assert lm_input.shape[1] == 282 and logps.shape[0] == 251
    rows = np.array(LLM_LONG_ROWS, dtype=np.int64)
    save("llm_full_long", text=text, prompt_text=prompt_text, prompt_speech=prompt_speech, forced=np.array(forced), rows=rows,
         lm_input_summary=summary(lm_input), logps=logps[torch.from_numpy(rows)])


def golden_sampler_ref():
    """The reference's OWN sampling functions (utils/common.py:105-146: ras_sampling, non_random_ras_sampling, nucleus_sampling,
    random_sampling) called as they are, with torch.Tensor.multinomial replaced by an inverse-CDF draw from recorded uniforms
    (torch.multinomial's stream cannot be reproduced on another device): what is pinned is every decision around the draw —
    the candidate set handed to multinomial, the repetition test, the fallback, the returned id."""
    from cosyvoice.utils import common as C
    g = torch.Generator().manual_seed(77)
    V, n_cases = 300, 48
    scores = torch.randn(n_cases, V, generator=g) * 2.5
    scores[::7] *= 0.2                                    # flat rows: the top-k cap binds before top-p
    uniforms = torch.rand(n_cases, 2, generator=g, dtype=torch.float64)
    uniforms[::3, 0] *= 0.05                              # rows with a forced repetition draw the top candidate: fallback taken
    hist = torch.randint(0, V, (n_cases, 12), generator=g)
    calls = []
    orig = torch.Tensor.multinomial

    def fake_multinomial(self, num_samples, replacement=False, *, generator=None):
        u = calls_state["u"][calls_state["k"]]
        calls_state["k"] += 1
        calls_state["inputs"].append(self.detach().clone())
        c = torch.cumsum(self.double() / self.double().sum(), 0)
        idx = int(torch.searchsorted(c, torch.tensor(u, dtype=torch.float64), right=True).item())
        return torch.tensor([min(idx, self.numel() - 1)])

    torch.Tensor.multinomial = fake_multinomial
    out = {"ras": [], "nrras": [], "ras_n1": [], "nrras_n1": [], "ras_n2": [], "nrras_n2": []}
    try:
        for name, fn, kw in (("ras", C.ras_sampling, {}), ("nrras", C.non_random_ras_sampling, dict(top_k=10, expand_scale=2))):
            for i in range(n_cases):
                dec = hist[i].tolist()
                if i % 3 == 0:      # force a repetition: the most likely id fills the window
                    dec = dec[:2] + [int(scores[i].argmax())] * 10
                calls_state = {"u": uniforms[i].tolist(), "k": 0, "inputs": []}
                tid = fn(scores[i], dec, 25, **kw)
                out[name].append(int(tid))
                out[name + "_n1"].append(calls_state["inputs"][0].numel())
                out[name + "_n2"].append(calls_state["inputs"][1].numel() if calls_state["k"] > 1 else 0)
    finally:
        torch.Tensor.multinomial = orig
    dec_all = []
    for i in range(n_cases):
        dec = hist[i].tolist()
        if i % 3 == 0:
            dec = dec[:2] + [int(scores[i].argmax())] * 10
        dec_all.append(dec)
    save("sampler_ref", scores=scores, uniforms=uniforms, decoded=np.array(dec_all, dtype=np.int64),
         **{k: np.array(v, dtype=np.int64) for k, v in out.items()})


def golden_flow_full():
    """FULL-depth reference flow (56 transformer blocks x 10 Euler steps, 6 + 4 conformer layers) on key-seeded weights at
    T = 100 (prompt 15 + 35 tokens) and T = 500 (prompt 75 + 175 tokens: BASELINE C1's 3 s prompt), both with the encoder chunk
    mask of CosyVoice2Model (static_chunk_size 50, cli/model.py:314) and with full attention (CosyVoiceModel wiring, :49-50)."""
    from cosyvoice_amd.config import FlowConfig
    from cosyvoice_amd.weights import flow_state_dict
    cfg = FlowConfig.full()
    flow = build_ref_flow(cfg, flow_state_dict(cfg))
    g = torch.Generator().manual_seed(61)
    out = {}
    for tag, n_p, n_g in (("t100", 15, 35), ("t500", 75, 175)):
        token = torch.randint(0, cfg.vocab_size, (1, n_g), generator=g, dtype=torch.int32)
        prompt_token = torch.randint(0, cfg.vocab_size, (1, n_p), generator=g, dtype=torch.int32)
        prompt_feat = torch.clamp(torch.randn(1, 2 * n_p, 80, generator=g) * 2 - 6, -11.5, 2.0)
        embedding = torch.randn(1, cfg.spk_embed_dim, generator=g)
        out.update({f"{tag}_token": token, f"{tag}_prompt_token": prompt_token, f"{tag}_prompt_feat": prompt_feat, f"{tag}_embedding": embedding})
        for ctag, chunk in (("chunk50", 50), ("full", 0)):
            flow.encoder.static_chunk_size = chunk
            with torch.inference_mode():
                mel, _ = flow.inference(token=token, token_len=torch.tensor([n_g]), prompt_token=prompt_token,
                                        prompt_token_len=torch.tensor([n_p]), prompt_feat=prompt_feat,
                                        prompt_feat_len=torch.tensor([2 * n_p]), embedding=embedding)
            out[f"{tag}_mel_{ctag}"] = mel
            out[f"{tag}_mel_{ctag}_chan_absmean"] = mel[0].abs().mean(dim=1)
            print(tag, ctag, tuple(mel.shape), summary(mel))
    save("flow_full", **out)


def golden_flow_long():
    """FULL-depth reference flow at the length of the C4 bench line: N_p = 250 prompt tokens + N_g = 250 generated tokens -> T = 1000
    frames through CausalMaskedDiffWithXvec.inference (flow/flow.py:258-319), encoder chunk mask 50 (CosyVoice2Model, cli/model.py:314),
    56 estimator blocks x 10 CFG Euler steps.  The returned mel is (1, 80, 500); the fixture keeps the exact fp32 mel on every 4th
    frame plus the per-channel abs-means and per-frame means of the whole mel."""
    from cosyvoice_amd.config import FlowConfig
    from cosyvoice_amd.weights import flow_state_dict
    cfg = FlowConfig.full()
    flow = build_ref_flow(cfg, flow_state_dict(cfg))
    g = torch.Generator().manual_seed(67)
    n_p, n_g = 250, 250
    token = torch.randint(0, cfg.vocab_size, (1, n_g), generator=g, dtype=torch.int32)
    prompt_token = torch.randint(0, cfg.vocab_size, (1, n_p), generator=g, dtype=torch.int32)
    prompt_feat = torch.clamp(torch.randn(1, 2 * n_p, 80, generator=g) * 2 - 6, -11.5, 2.0)
    embedding = torch.randn(1, cfg.spk_embed_dim, generator=g)
    flow.encoder.static_chunk_size = 50
    with torch.inference_mode():
        mel, _ = flow.inference(token=token, token_len=torch.tensor([n_g]), prompt_token=prompt_token,
                                prompt_token_len=torch.tensor([n_p]), prompt_feat=prompt_feat,
                                prompt_feat_len=torch.tensor([2 * n_p]), embedding=embedding)
    assert mel.shape == (1, 80, 2 * n_g), mel.shape
    print("flow long", tuple(mel.shape), summary(mel))
    save("flow_long", token=token, prompt_token=prompt_token, prompt_feat=prompt_feat, embedding=embedding,
         mel_sub4=mel[:, :, ::4].contiguous(), mel_chan_absmean=mel[0].abs().mean(dim=1), mel_frame_mean=mel[0].mean(dim=0),
         mel_summary=summary(mel))


def golden_hift_long():
    """Reference HiFTGenerator.decode (hifigan/generator.py:349-381) at the BASELINE lengths: v2 (24 kHz) 500 frames -> 240 000 samples,
    v1 (22.05 kHz) 861 frames -> 220 416 samples, injected source s.  The waveform is kept on every 8th sample (exact fp32) plus
    per-2400-sample block abs-means / abs-maxima of the whole waveform; the inputs are regenerated from their seeds by the test."""
    from cosyvoice_amd.config import HiftConfig
    for tag, cfg, frames in (("v2", HiftConfig.v2(), 500), ("v1", HiftConfig.v1(), 861)):
        from cosyvoice_amd.weights import hift_state_dict
        m = build_ref_hift(cfg, hift_state_dict(cfg))
        mel = synth_mel(1, frames, seed=211)
        g = torch.Generator().manual_seed(212)
        s = torch.randn(1, 1, frames * cfg.total_upsample, generator=g) * 0.05
        with torch.inference_mode():
            wav = m.decode(x=mel, s=s)
            f0 = m.f0_predictor(mel)
        n = wav.shape[1]
        blk = wav[0, : n // 2400 * 2400].view(-1, 2400).abs() if n >= 2400 else wav.abs()
        print("hift long", tag, tuple(wav.shape), summary(wav))
        save(f"hift_{tag}_long", frames=np.array(frames), mel_seed=np.array(211), s_seed=np.array(212), n_samples=np.array(n),
             wav_sub8=wav[:, ::8].contiguous(), wav_block_absmean=blk.mean(dim=1), wav_block_absmax=blk.max(dim=1).values,
             f0=f0, wav_summary=summary(wav))


def golden_stream_v2():
    """The reference's own CosyVoice2Model (cli/model.py:295-424) streaming path: tts(stream=True) with a stub LLM that emits a
    fixed token list and a stub vocoder that records the speech_feat / cache_source it is handed (the real HiFT draws random
    source phases: its waveform is not comparable) and returns a deterministic "waveform" (each mel frame's mean repeated 480
    times), so that the mel / source / speech caches, token_offset trimming and the hamming cross-fade of token2wav (:334-366)
    all run as written.  With rand_noise fixed (flow_matching.py:212-213) every chunk's mel is deterministic."""
    import threading  # noqa: F401
    from cosyvoice.cli import model as ref_model
    from cosyvoice_amd.config import FlowConfig
    from cosyvoice_amd.weights import flow_state_dict
    fc = FlowConfig.tiny()
    flow = build_ref_flow(fc, flow_state_dict(fc))
    g = torch.Generator().manual_seed(123)
    n_tok, n_p = 137, 12                      # 137 = 2 full hops of 50 + 3 look-ahead ... + a 37-token tail
    tokens = torch.randint(0, fc.vocab_size, (n_tok,), generator=g).tolist()
    ptok = torch.randint(0, fc.vocab_size, (1, n_p), generator=g, dtype=torch.int32)
    pfeat = torch.clamp(torch.randn(1, 2 * n_p, 80, generator=g) * 2 - 6, -11.5, 2.0)
    emb = torch.randn(1, fc.spk_embed_dim, generator=g)

    class StubLLM(torch.nn.Module):
        fp16 = False

        def inference(self, **kw):
            for t in tokens:
                yield t

    calls = []

    class StubHift(torch.nn.Module):
        def inference(self, speech_feat, cache_source=torch.zeros(1, 1, 0)):
            calls.append((speech_feat.detach().clone(), cache_source.detach().clone()))
            wav = speech_feat.mean(dim=1).repeat_interleave(480, dim=1)           # (1, T * 480)
            src = wav.unsqueeze(1) * 0.5
            if cache_source.shape[2] != 0:                                        # generator.py:408-409
                src[:, :, :cache_source.shape[2]] = cache_source
            return wav, src

    m = ref_model.CosyVoice2Model(StubLLM(), flow, StubHift(), fp16=False)
    m.device = torch.device("cpu")
    from contextlib import nullcontext
    m.llm_context = nullcontext()
    with torch.inference_mode():
        chunks = [o["tts_speech"] for o in m.tts(text=torch.zeros(1, 5, dtype=torch.int32), flow_embedding=emb,
                                                  flow_prompt_speech_token=ptok, prompt_speech_feat=pfeat, stream=True)]
    out = {"tokens": np.array(tokens, dtype=np.int32), "prompt_token": ptok, "prompt_feat": pfeat, "embedding": emb,
           "chunk_samples": np.array([c.shape[1] for c in chunks], dtype=np.int64), "n_calls": np.array(len(calls))}
    for i, ((feat, src), c) in enumerate(zip(calls, chunks)):
        out[f"feat{i}"] = feat          # the mel handed to the vocoder (cache frames prepended, token_offset trimmed)
        out[f"src{i}"] = src[:, :, ::480] if src.shape[2] else src   # cache_source, one sample per frame (it is piecewise constant)
        out[f"wav{i}"] = c[:, ::160]    # the yielded chunk after the cross-fade, decimated (keeps the fixture small)
    print("stream chunks", out["chunk_samples"], "vocoder calls", len(calls))
    save("stream_v2", **out)



def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    install_stubs()
    which = sys.argv[1:] or ["hift", "flow", "llm", "bigvgan", "frontend", "phoneme", "v1orch", "llmv1", "flowv1", "llmloop", "samplerref",
                             "flowfull", "streamv2", "llmlong", "flowlong", "hiftlong"]
    if "hift" in which:
        golden_hift()
    if "flow" in which:
        golden_flow()
    if "llm" in which:
        golden_llm()
    if "bigvgan" in which:
        golden_bigvgan_act()
        golden_bigvgan_model()
    if "frontend" in which:
        golden_frontend_mel()
    if "phoneme" in which:
        golden_llm_phoneme()
    if "v1orch" in which:
        golden_v1_orchestrator()
    if "llmv1" in which:
        golden_llm_v1()
    if "flowv1" in which:
        golden_flow_v1()
    if "llmloop" in which:
        golden_llm_loop()
    if "samplerref" in which:
        golden_sampler_ref()
    if "flowfull" in which:
        golden_flow_full()
    if "streamv2" in which:
        golden_stream_v2()
    if "llmlong" in which:
        golden_llm_long()
    if "flowlong" in which:
        golden_flow_long()
    if "hiftlong" in which:
        golden_hift_long()


if __name__ == "__main__":
    main()
