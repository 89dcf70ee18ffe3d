"""Deterministic, key-name-seeded synthetic state dicts with the reference's key names.

No checkpoint exists in the build environment, so parity and the
benchmark run on synthetic weights.  Every tensor is drawn from a CPU
``torch.Generator`` seeded by a hash of its state-dict key, so the same bytes are
regenerated on any machine (here to mint goldens through the reference's own
``load_state_dict``; on the GPU box to feed the HIP path and the oracle).

Key names follow the reference module tree (SURVEY.md §8b "State-dict contract"):
  hift : /root/reference/cosyvoice/hifigan/generator.py:268-316, f0_predictor.py:27-50
  flow : /root/reference/cosyvoice/flow/flow.py:195-201, flow/decoder.py:110-206,
         transformer/upsample_encoder.py:177-235, transformer/attention.py:44-48,216-221
  llm  : /root/reference/cosyvoice/llm/llm.py:786-801 + HF Qwen2 module names
"""
import hashlib
import math
from typing import Dict, List, Tuple

import torch

from .config import FlowConfig, HiftConfig, LlmConfig

Spec = Tuple[str, Tuple[int, ...], str, float]  # key, shape, kind, param


def _seed(key: str, base: int) -> int:
    h = hashlib.sha256(f"{base}:{key}".encode()).digest()
    return int.from_bytes(h[:8], "little") & 0x7FFFFFFFFFFFFFFF


def _draw(key: str, shape, kind: str, param: float, base: int) -> torch.Tensor:
    g = torch.Generator(device="cpu")
    g.manual_seed(_seed(key, base))
    if kind == "normal":
        return torch.randn(shape, generator=g, dtype=torch.float32) * param
    if kind == "uniform":  # U(-param, param)
        return (torch.rand(shape, generator=g, dtype=torch.float32) * 2 - 1) * param
    if kind == "norm_w":  # LayerNorm / RMSNorm gain
        return 1.0 + torch.randn(shape, generator=g, dtype=torch.float32) * param
    if kind == "alpha":  # snake alpha in (0.5, 1.5)
        return 0.5 + torch.rand(shape, generator=g, dtype=torch.float32)
    if kind == "const":
        return torch.full(shape, param, dtype=torch.float32)
    raise ValueError(kind)


def materialize(specs: List[Spec], base_seed: int = 1986, round_to: torch.dtype = None) -> Dict[str, torch.Tensor]:
    """Draw every tensor of ``specs``.  ``weight_g`` entries (kind 'wn_g') are derived
    from the already-drawn ``weight_v`` so that g = ||v|| * U(0.7, 1.3) exercises the
    legacy weight-norm fold."""
    sd: Dict[str, torch.Tensor] = {}
    for key, shape, kind, param in specs:
        if kind == "wn_g":
            continue
        sd[key] = _draw(key, shape, kind, param, base_seed)
    for key, shape, kind, param in specs:
        if kind != "wn_g":
            continue
        v = sd[key[:-1] + "v"]
        nrm = v.reshape(v.shape[0], -1).norm(dim=1).reshape(shape)
        g = torch.Generator(device="cpu")
        g.manual_seed(_seed(key, base_seed))
        sd[key] = nrm * (0.7 + 0.6 * torch.rand(shape, generator=g, dtype=torch.float32))
    if round_to is not None:
        for k in sd:
            sd[k] = sd[k].to(round_to).to(torch.float32)
    return sd


# --------------------------------------------------------------------------- HiFT
def _wn_conv(specs, name, cout, cin, k, std, transpose=False, bias_std=0.02):
    shape = (cin, cout, k) if transpose else (cout, cin, k)
    d0 = shape[0]
    specs.append((f"{name}.weight_v", shape, "normal", std))
    specs.append((f"{name}.weight_g", (d0, 1, 1), "wn_g", 0.0))
    specs.append((f"{name}.bias", (cout,), "normal", bias_std))


def _resblock(specs, name, ch, k, dils):
    for j in range(len(dils)):
        _wn_conv(specs, f"{name}.convs1.{j}", ch, ch, k, 0.01)
        _wn_conv(specs, f"{name}.convs2.{j}", ch, ch, k, 0.01)
    for j in range(len(dils)):
        specs.append((f"{name}.activations1.{j}.alpha", (ch,), "alpha", 0.0))
        specs.append((f"{name}.activations2.{j}.alpha", (ch,), "alpha", 0.0))


def hift_downsample_plan(cfg: HiftConfig):
    """(stride u, kernel, padding) of each source_downs conv — generator.py:289-300."""
    ups = list(cfg.upsample_rates)
    down = [1] + ups[::-1][:-1]
    cum = []
    c = 1
    for d in down:
        c *= d
        cum.append(c)
    plan = []
    for u in cum[::-1]:
        if u == 1:
            plan.append((1, 1, 0))
        else:
            plan.append((u, u * 2, u // 2))
    return plan


def hift_specs(cfg: HiftConfig) -> List[Spec]:
    s: List[Spec] = []
    nh = cfg.nb_harmonics + 1
    s.append(("m_source.l_linear.weight", (1, nh), "uniform", 1.0 / math.sqrt(nh)))
    s.append(("m_source.l_linear.bias", (1,), "uniform", 1.0 / math.sqrt(nh)))
    bc = cfg.base_channels
    _wn_conv(s, "conv_pre", bc, cfg.in_channels, 7, 1.0 / math.sqrt(cfg.in_channels * 7 * 3))
    for i, (u, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
        _wn_conv(s, f"ups.{i}", bc // 2 ** (i + 1), bc // 2 ** i, k, 0.01, transpose=True)
    nfft2 = cfg.n_fft + 2
    for i, (stride, k, pad) in enumerate(hift_downsample_plan(cfg)):
        ch = bc // 2 ** (i + 1)
        bound = 1.0 / math.sqrt(nfft2 * k)
        s.append((f"source_downs.{i}.weight", (ch, nfft2, k), "uniform", bound))
        s.append((f"source_downs.{i}.bias", (ch,), "uniform", bound))
        _resblock(s, f"source_resblocks.{i}", ch, cfg.source_resblock_kernel_sizes[i],
                  cfg.source_resblock_dilation_sizes[i])
    nk = len(cfg.resblock_kernel_sizes)
    ch = bc
    for i in range(len(cfg.upsample_rates)):
        ch = bc // 2 ** (i + 1)
        for j, (k, d) in enumerate(zip(cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes)):
            _resblock(s, f"resblocks.{i * nk + j}", ch, k, d)
    # conv_post: magnitude-channel bias pulled negative so exp() keeps the synthetic waveform unclipped
    s.append(("conv_post.weight_v", (nfft2, ch, 7), "normal", 0.01))
    s.append(("conv_post.weight_g", (nfft2, 1, 1), "wn_g", 0.0))
    s.append(("conv_post.bias", (nfft2,), "normal", 0.3))
    fc = cfg.f0_cond_channels
    cin = cfg.in_channels
    for idx in (0, 2, 4, 6, 8):
        _wn_conv(s, f"f0_predictor.condnet.{idx}", fc, cin, 3, 1.0 / math.sqrt(cin * 3))
        cin = fc
    # classifier scaled so the synthetic F0 spans voiced/unvoiced around the 10 Hz threshold
    s.append(("f0_predictor.classifier.weight", (1, fc), "normal", 40.0 / math.sqrt(fc)))
    s.append(("f0_predictor.classifier.bias", (1,), "const", 60.0))
    return s


# --------------------------------------------------------------------------- flow
def _linear(specs, name, out_f, in_f, bias=True, gain=1.0, bias_std=0.02):
    specs.append((f"{name}.weight", (out_f, in_f), "normal", gain / math.sqrt(in_f)))
    if bias:
        specs.append((f"{name}.bias", (out_f,), "normal", bias_std))


def _conv(specs, name, cout, cin, k, gain=1.0, bias_std=0.02):
    specs.append((f"{name}.weight", (cout, cin, k), "normal", gain / math.sqrt(cin * k)))
    specs.append((f"{name}.bias", (cout,), "normal", bias_std))


def _ln(specs, name, dim):
    specs.append((f"{name}.weight", (dim,), "norm_w", 0.1))
    specs.append((f"{name}.bias", (dim,), "normal", 0.05))


def _conformer_layer(specs, name, dim, heads, units):
    dk = dim // heads
    for n in ("linear_q", "linear_k", "linear_v", "linear_out"):
        _linear(specs, f"{name}.self_attn.{n}", dim, dim)
    _linear(specs, f"{name}.self_attn.linear_pos", dim, dim, bias=False)
    bound = math.sqrt(6.0 / (heads + dk))
    specs.append((f"{name}.self_attn.pos_bias_u", (heads, dk), "uniform", bound))
    specs.append((f"{name}.self_attn.pos_bias_v", (heads, dk), "uniform", bound))
    _linear(specs, f"{name}.feed_forward.w_1", units, dim)
    _linear(specs, f"{name}.feed_forward.w_2", dim, units)
    _ln(specs, f"{name}.norm_ff", dim)
    _ln(specs, f"{name}.norm_mha", dim)


def _est_resnet(specs, name, cin, cout, tdim, norm_idx=2):
    for b, ci in (("block1", cin), ("block2", cout)):
        _conv(specs, f"{name}.{b}.block.0", cout, ci, 3)
        _ln(specs, f"{name}.{b}.block.{norm_idx}", cout)
    _linear(specs, f"{name}.mlp.1", cout, tdim)
    _conv(specs, f"{name}.res_conv", cout, cin, 1)


def _est_tblock(specs, name, dim, inner, ff):
    _ln(specs, f"{name}.norm1", dim)
    for n in ("to_q", "to_k", "to_v"):
        _linear(specs, f"{name}.attn1.{n}", inner, dim, bias=False)
    _linear(specs, f"{name}.attn1.to_out.0", dim, inner, gain=0.5)
    _ln(specs, f"{name}.norm3", dim)
    _linear(specs, f"{name}.ff.net.0.proj", ff, dim)
    _linear(specs, f"{name}.ff.net.2", dim, ff, gain=0.5)


def estimator_specs(cfg: FlowConfig, prefix: str = "decoder.estimator.") -> List[Spec]:
    s: List[Spec] = []
    C, tdim = cfg.est_channels, cfg.est_time_dim
    inner, ff = cfg.est_inner, cfg.est_channels * cfg.est_ff_mult
    _linear(s, f"{prefix}time_mlp.linear_1", tdim, cfg.est_in_channels)
    _linear(s, f"{prefix}time_mlp.linear_2", tdim, tdim)
    _est_resnet(s, f"{prefix}down_blocks.0.0", cfg.est_in_channels, C, tdim)
    for j in range(cfg.est_n_blocks):
        _est_tblock(s, f"{prefix}down_blocks.0.1.{j}", C, inner, ff)
    _conv(s, f"{prefix}down_blocks.0.2", C, C, 3)
    for i in range(cfg.est_mid_blocks):
        _est_resnet(s, f"{prefix}mid_blocks.{i}.0", C, C, tdim)
        for j in range(cfg.est_n_blocks):
            _est_tblock(s, f"{prefix}mid_blocks.{i}.1.{j}", C, inner, ff)
    _est_resnet(s, f"{prefix}up_blocks.0.0", 2 * C, C, tdim)
    for j in range(cfg.est_n_blocks):
        _est_tblock(s, f"{prefix}up_blocks.0.1.{j}", C, inner, ff)
    _conv(s, f"{prefix}up_blocks.0.2", C, C, 3)
    _conv(s, f"{prefix}final_block.block.0", C, C, 3)
    _ln(s, f"{prefix}final_block.block.2", C)
    _conv(s, f"{prefix}final_proj", cfg.output_size, C, 1)
    return s


def encoder_specs(cfg: FlowConfig, prefix: str = "encoder.") -> List[Spec]:
    s: List[Spec] = []
    D = cfg.enc_dim
    _linear(s, f"{prefix}embed.out.0", D, cfg.input_size)
    _ln(s, f"{prefix}embed.out.1", D)
    _ln(s, f"{prefix}after_norm", D)
    _conv(s, f"{prefix}pre_lookahead_layer.conv1", D, D, cfg.pre_lookahead_len + 1)
    _conv(s, f"{prefix}pre_lookahead_layer.conv2", D, D, 3)
    for i in range(cfg.enc_blocks):
        _conformer_layer(s, f"{prefix}encoders.{i}", D, cfg.enc_heads, cfg.enc_linear_units)
    _conv(s, f"{prefix}up_layer.conv", D, D, 5)
    _linear(s, f"{prefix}up_embed.out.0", D, cfg.input_size)
    _ln(s, f"{prefix}up_embed.out.1", D)
    for i in range(cfg.enc_up_blocks):
        _conformer_layer(s, f"{prefix}up_encoders.{i}", D, cfg.enc_heads, cfg.enc_linear_units)
    return s


def flow_specs(cfg: FlowConfig) -> List[Spec]:
    s: List[Spec] = []
    s.append(("input_embedding.weight", (cfg.vocab_size, cfg.input_size), "normal", 1.0))
    _linear(s, "spk_embed_affine_layer", cfg.output_size, cfg.spk_embed_dim, gain=3.0)
    s += encoder_specs(cfg)
    _linear(s, "encoder_proj", cfg.output_size, cfg.enc_dim)
    s += estimator_specs(cfg)
    return s


# --------------------------------------------------------------------------- llm
def llm_specs(cfg: LlmConfig) -> List[Spec]:
    s: List[Spec] = []
    H, I = cfg.hidden_size, cfg.intermediate_size
    p = "llm.model.model."
    s.append((f"{p}embed_tokens.weight", (cfg.vocab_size, H), "normal", 0.02))
    for i in range(cfg.num_layers):
        lp = f"{p}layers.{i}."
        s.append((f"{lp}self_attn.q_proj.weight", (cfg.q_dim, H), "normal", 1.0 / math.sqrt(H)))
        s.append((f"{lp}self_attn.q_proj.bias", (cfg.q_dim,), "normal", 0.1))
        s.append((f"{lp}self_attn.k_proj.weight", (cfg.kv_dim, H), "normal", 1.0 / math.sqrt(H)))
        s.append((f"{lp}self_attn.k_proj.bias", (cfg.kv_dim,), "normal", 0.1))
        s.append((f"{lp}self_attn.v_proj.weight", (cfg.kv_dim, H), "normal", 1.0 / math.sqrt(H)))
        s.append((f"{lp}self_attn.v_proj.bias", (cfg.kv_dim,), "normal", 0.1))
        s.append((f"{lp}self_attn.o_proj.weight", (H, cfg.q_dim), "normal", 0.5 / math.sqrt(cfg.q_dim)))
        s.append((f"{lp}mlp.gate_proj.weight", (I, H), "normal", 1.0 / math.sqrt(H)))
        s.append((f"{lp}mlp.up_proj.weight", (I, H), "normal", 1.0 / math.sqrt(H)))
        s.append((f"{lp}mlp.down_proj.weight", (H, I), "normal", 0.5 / math.sqrt(I)))
        s.append((f"{lp}input_layernorm.weight", (H,), "norm_w", 0.1))
        s.append((f"{lp}post_attention_layernorm.weight", (H,), "norm_w", 0.1))
    s.append((f"{p}norm.weight", (H,), "norm_w", 0.1))
    if not cfg.tie_word_embeddings:
        s.append(("llm.model.lm_head.weight", (cfg.vocab_size, H), "normal", 0.02))
    s.append(("llm_embedding.weight", (2, H), "normal", 0.02))
    s.append(("llm_decoder.weight", (cfg.out_vocab, H), "normal", 2.0 / math.sqrt(H)))
    s.append(("llm_decoder.bias", (cfg.out_vocab,), "normal", 0.1))
    s.append(("speech_embedding.weight", (cfg.out_vocab, H), "normal", 0.02))
    return s


def hift_state_dict(cfg: HiftConfig, seed: int = 1986):
    return materialize(hift_specs(cfg), seed)


def flow_state_dict(cfg: FlowConfig, seed: int = 1986, round_to=None):
    return materialize(flow_specs(cfg), seed, round_to)


def llm_state_dict(cfg: LlmConfig, seed: int = 1986, round_to=None):
    sd = materialize(llm_specs(cfg), seed, round_to)
    if cfg.tie_word_embeddings:
        sd["llm.model.lm_head.weight"] = sd["llm.model.model.embed_tokens.weight"]
    return sd


def fold_weight_norm(sd: Dict[str, torch.Tensor], name: str) -> torch.Tensor:
    """w = g * v / ||v||, norm over dims (1,2) per index of dim 0 (legacy torch weight_norm, dim=0);
    for ConvTranspose1d that is per *input* channel."""
    if f"{name}.weight" in sd:
        return sd[f"{name}.weight"]
    v, g = sd[f"{name}.weight_v"], sd[f"{name}.weight_g"]
    nrm = v.reshape(v.shape[0], -1).norm(dim=1).reshape(g.shape)
    return v * (g / nrm)


# --------------------------------------------------------------------------- BigVGAN
def bigvgan_specs(cfg) -> List[Spec]:
    """Keys of the reference's BigVGAN module tree (BigVGAN/bigvgan.py:257-382) for encoder1 = encoder2 = None."""
    s: List[Spec] = []
    s.append(("input_embedding.weight", (cfg.vocab_size, cfg.input_size), "normal", 1.0))
    _linear(s, "encoder_proj", cfg.output_size, cfg.input_size)
    c0 = cfg.upsample_initial_channel
    _linear(s, "mel_proj", cfg.mel_bin, c0)
    _wn_conv(s, "conv_pre", c0, cfg.output_size, 7, 1.0 / math.sqrt(cfg.output_size * 7))
    sd = cfg.speaker_embedding_dim
    s.append(("cond_layer.weight", (c0, sd, 1), "normal", 0.5 / math.sqrt(sd)))
    s.append(("cond_layer.bias", (c0,), "normal", 0.02))
    nk = len(cfg.resblock_kernel_sizes)
    ch = c0
    for i, (u, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
        ch = c0 // 2 ** (i + 1)
        _wn_conv(s, f"ups.{i}.0", ch, c0 // 2 ** i, k, 1.0 / math.sqrt(c0 // 2 ** i * k / u), transpose=True)
        if cfg.cond_in_each_up_layer:
            s.append((f"conds.{i}.weight", (ch, sd, 1), "normal", 0.5 / math.sqrt(sd)))
            s.append((f"conds.{i}.bias", (ch,), "normal", 0.02))
        for j, (k2, dils) in enumerate(zip(cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes)):
            name = f"resblocks.{i * nk + j}"
            for d in range(len(dils)):
                _wn_conv(s, f"{name}.convs1.{d}", ch, ch, k2, 0.5 / math.sqrt(ch * k2))
                _wn_conv(s, f"{name}.convs2.{d}", ch, ch, k2, 0.5 / math.sqrt(ch * k2))
            for m in range(2 * len(dils)):
                s.append((f"{name}.activations.{m}.act.alpha", (ch,), "normal", 0.3))   # log-scale parameters
                s.append((f"{name}.activations.{m}.act.beta", (ch,), "normal", 0.3))
    s.append(("activation_post.act.alpha", (ch,), "normal", 0.3))
    s.append(("activation_post.act.beta", (ch,), "normal", 0.3))
    _wn_conv(s, "conv_post", 1, ch, 7, 0.3 / math.sqrt(ch * 7))   # keeps tanh out of saturation on synthetic weights
    return s


def bigvgan_state_dict(cfg, seed: int = 0, round_to=None) -> Dict[str, torch.Tensor]:
    return materialize(bigvgan_specs(cfg), seed, round_to)


# --------------------------------------------------------------------------- Qwen2LM_Phoneme_Src2
def phoneme_lm_specs(pcfg, lcfg: LlmConfig) -> List[Spec]:
    """llm.py:1482-1531 module tree: text_embedding.{0..3}, text_encoder (ConformerEncoder), text_encoder_affine_layer,
    src_attention.0 (DecoderLayer), spk_embed_affine_layer + the Qwen2 stack / llm_embedding / llm_decoder / speech_embedding."""
    s: List[Spec] = llm_specs(lcfg)
    H, D = lcfg.hidden_size, pcfg.enc_dim
    for i, (n, d) in enumerate(((pcfg.text_token_size, pcfg.text_token_dim), (pcfg.text_tone_size, pcfg.text_tone_dim),
                                (pcfg.text_lang_size, pcfg.text_lang_dim), (pcfg.text_prsd_size, pcfg.text_prsd_dim))):
        s.append((f"text_embedding.{i}.weight", (n, d), "normal", 1.0))
    _linear(s, "text_encoder.embed.out.0", D, pcfg.input_size)
    _ln(s, "text_encoder.embed.out.1", D)
    _ln(s, "text_encoder.after_norm", D)
    for i in range(pcfg.enc_blocks):
        _conformer_layer(s, f"text_encoder.encoders.{i}", D, pcfg.enc_heads, pcfg.enc_linear_units)
    _linear(s, "text_encoder_affine_layer", H, D)
    for att in ("self_attn", "src_attn"):
        for lin in ("linear_q", "linear_k", "linear_v", "linear_out"):
            _linear(s, f"src_attention.0.{att}.{lin}", H, H, gain=(0.5 if lin == "linear_out" else 1.0))
    _linear(s, "src_attention.0.feed_forward.w_1", pcfg.src_linear_units, H)
    _linear(s, "src_attention.0.feed_forward.w_2", H, pcfg.src_linear_units, gain=0.5)
    for n in ("norm1", "norm2", "norm3"):
        _ln(s, f"src_attention.0.{n}", H)
    _linear(s, "spk_embed_affine_layer", H, pcfg.spk_embed_dim)
    return s


def phoneme_lm_state_dict(pcfg, lcfg: LlmConfig, seed: int = 1986, round_to=None):
    sd = materialize(phoneme_lm_specs(pcfg, lcfg), seed, round_to)
    if lcfg.tie_word_embeddings:
        sd["llm.model.lm_head.weight"] = sd["llm.model.model.embed_tokens.weight"]
    return sd


# --------------------------------------------------------------------------- CosyVoice-v1 TransformerLM
def _transformer_layer(specs, name, dim, heads, units):
    """TransformerEncoderLayer with rel-pos attention (encoder_layer.py:24-107): same tensors as a conformer layer, norm1 / norm2."""
    dk = dim // heads
    for n in ("linear_q", "linear_k", "linear_v", "linear_out"):
        _linear(specs, f"{name}.self_attn.{n}", dim, dim)
    _linear(specs, f"{name}.self_attn.linear_pos", dim, dim, bias=False)
    bound = math.sqrt(6.0 / (heads + dk))
    specs.append((f"{name}.self_attn.pos_bias_u", (heads, dk), "uniform", bound))
    specs.append((f"{name}.self_attn.pos_bias_v", (heads, dk), "uniform", bound))
    _linear(specs, f"{name}.feed_forward.w_1", units, dim)
    _linear(specs, f"{name}.feed_forward.w_2", dim, units)
    _ln(specs, f"{name}.norm1", dim)
    _ln(specs, f"{name}.norm2", dim)


def transformer_lm_specs(cfg) -> List[Spec]:
    s: List[Spec] = []
    s.append(("text_embedding.weight", (cfg.text_token_size, cfg.text_encoder_input_size), "normal", 1.0))
    _linear(s, "text_encoder.embed.out.0", cfg.enc_dim, cfg.text_encoder_input_size)
    _ln(s, "text_encoder.embed.out.1", cfg.enc_dim)
    _ln(s, "text_encoder.after_norm", cfg.enc_dim)
    for i in range(cfg.enc_blocks):
        _conformer_layer(s, f"text_encoder.encoders.{i}", cfg.enc_dim, cfg.enc_heads, cfg.enc_linear_units)
    _linear(s, "text_encoder_affine_layer", cfg.llm_dim, cfg.enc_dim)
    s.append(("llm_embedding.weight", (2, cfg.llm_dim), "normal", 1.0))
    _linear(s, "llm.embed.out.0", cfg.llm_dim, cfg.llm_dim)
    _ln(s, "llm.embed.out.1", cfg.llm_dim)
    _ln(s, "llm.after_norm", cfg.llm_dim)
    for i in range(cfg.llm_blocks):
        _transformer_layer(s, f"llm.encoders.{i}", cfg.llm_dim, cfg.llm_heads, cfg.llm_linear_units)
    _linear(s, "llm_decoder", cfg.speech_token_size + 1, cfg.llm_dim, gain=2.0, bias_std=0.1)
    s.append(("speech_embedding.weight", (cfg.speech_token_size, cfg.llm_dim), "normal", 1.0))
    _linear(s, "spk_embed_affine_layer", cfg.llm_dim, cfg.spk_embed_dim)
    return s


def transformer_lm_state_dict(cfg, seed: int = 1986, round_to=None):
    return materialize(transformer_lm_specs(cfg), seed, round_to)


# --------------------------------------------------------------------------- CosyVoice-v1 flow (MaskedDiffWithXvec)
def estimator_v1_specs(cfg, prefix: str = "decoder.estimator.") -> List[Spec]:
    """Non-causal ConditionalDecoder channels=[C, C] (flow/decoder.py:110-206): down_blocks.0 ends in Downsample1D (`.2.conv`,
    k3 stride 2), down_blocks.1 in a plain Conv1d, up_blocks.0 in Upsample1D (`.2.conv`, ConvTranspose1d(4,2,1)), up_blocks.1
    in a plain Conv1d; every Block1D norm is a GroupNorm at `.block.1` (Conv1d, GroupNorm, Mish; the causal block has a Transpose in between)."""
    s: List[Spec] = []
    C, tdim = cfg.est_channels, cfg.est_time_dim
    inner, ff = cfg.est_inner, cfg.est_channels * cfg.est_ff_mult
    _linear(s, f"{prefix}time_mlp.linear_1", tdim, cfg.est_in_channels)
    _linear(s, f"{prefix}time_mlp.linear_2", tdim, tdim)

    def stage(name, cin):
        _est_resnet(s, f"{name}.0", cin, C, tdim, norm_idx=1)
        for j in range(cfg.est_n_blocks):
            _est_tblock(s, f"{name}.1.{j}", C, inner, ff)

    stage(f"{prefix}down_blocks.0", cfg.est_in_channels)
    _conv(s, f"{prefix}down_blocks.0.2.conv", C, C, 3)
    stage(f"{prefix}down_blocks.1", C)
    _conv(s, f"{prefix}down_blocks.1.2", C, C, 3)
    for i in range(cfg.est_mid_blocks):
        stage(f"{prefix}mid_blocks.{i}", C)
    stage(f"{prefix}up_blocks.0", 2 * C)
    s.append((f"{prefix}up_blocks.0.2.conv.weight", (C, C, 4), "normal", 1.0 / math.sqrt(C * 2)))
    s.append((f"{prefix}up_blocks.0.2.conv.bias", (C,), "normal", 0.02))
    stage(f"{prefix}up_blocks.1", 2 * C)
    _conv(s, f"{prefix}up_blocks.1.2", C, C, 3)
    _conv(s, f"{prefix}final_block.block.0", C, C, 3)
    _ln(s, f"{prefix}final_block.block.1", C)
    _conv(s, f"{prefix}final_proj", cfg.output_size, C, 1)
    return s


def flow_v1_specs(cfg) -> List[Spec]:
    """flow/flow.py:47-62 module tree: input_embedding, spk_embed_affine_layer, encoder (ConformerEncoder), encoder_proj,
    length_regulator.model (Conv1d k3, GroupNorm, Mish) x reg_layers + Conv1d k1, decoder.estimator."""
    s: List[Spec] = []
    D, O = cfg.enc_dim, cfg.output_size
    s.append(("input_embedding.weight", (cfg.vocab_size, cfg.input_size), "normal", 1.0))
    _linear(s, "spk_embed_affine_layer", O, cfg.spk_embed_dim, gain=3.0)
    _linear(s, "encoder.embed.out.0", D, cfg.input_size)
    _ln(s, "encoder.embed.out.1", D)
    _ln(s, "encoder.after_norm", D)
    for i in range(cfg.enc_blocks):
        _conformer_layer(s, f"encoder.encoders.{i}", D, cfg.enc_heads, cfg.enc_linear_units)
    _linear(s, "encoder_proj", O, D)
    for i in range(cfg.reg_layers):
        _conv(s, f"length_regulator.model.{3 * i}", O, O, 3)
        _ln(s, f"length_regulator.model.{3 * i + 1}", O)
    _conv(s, f"length_regulator.model.{3 * cfg.reg_layers}", O, O, 1)
    s += estimator_v1_specs(cfg)
    return s


def flow_v1_state_dict(cfg, seed: int = 1986, round_to=None):
    return materialize(flow_v1_specs(cfg), seed, round_to)
