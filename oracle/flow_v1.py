"""Oracle for the CosyVoice-v1 flow ``MaskedDiffWithXvec`` (TEST INFRASTRUCTURE — see oracle/__init__.py).

Restates /root/reference/cosyvoice/flow/flow.py:108-160 (inference), flow/length_regulator.py:49-70
(InterpolateRegulator.inference), flow/flow_matching.py:37-70 (ConditionalCFM.forward with the flow cache) and the
non-causal two-level ConditionalDecoder (flow/decoder.py:222-334 with flow/components/decoder.py:30-68,118-156).
Pinned by tests/golden/flow_v1_tiny.npz, minted from the reference modules themselves (make_golden.golden_flow_v1);
the estimator's transformer blocks go through the same restated diffusers-0.27.2 classes as the CosyVoice2 estimator
golden (parity unpinned at that third-party boundary).  Batch 1 (the reference asserts it), so every mask
is all ones and is omitted."""
import torch
import torch.nn.functional as F

from . import flow as of


def _gn(sd, name, x, groups):
    return F.group_norm(x, groups, sd[f"{name}.weight"], sd[f"{name}.bias"], 1e-5)


def block1d(sd, name, x, groups):
    # Block1D, flow/components/decoder.py:30-41: Conv1d k3 pad 1 -> GroupNorm -> Mish
    h = F.conv1d(x, sd[f"{name}.block.0.weight"], sd[f"{name}.block.0.bias"], padding=1)
    return F.mish(_gn(sd, f"{name}.block.1", h, groups))


def resnet_block(sd, name, x, temb, groups):
    # ResnetBlock1D.forward, flow/components/decoder.py:54-59
    h = block1d(sd, f"{name}.block1", x, groups)
    h = h + of._lin(sd, f"{name}.mlp.1", F.mish(temb)).unsqueeze(-1)
    h = block1d(sd, f"{name}.block2", h, groups)
    return h + F.conv1d(x, sd[f"{name}.res_conv.weight"], sd[f"{name}.res_conv.bias"])


def estimator_forward(sd, cfg, x, mu, t, spks, cond, prefix="decoder.estimator."):
    """ConditionalDecoder.forward (causal=False, channels=[C, C]); x, mu, cond (B,80,T); t (B,); spks (B,80) -> (B,80,T)."""
    temb = of.sinusoidal_pos_emb(t, cfg.est_in_channels)
    temb = of._lin(sd, f"{prefix}time_mlp.linear_2", F.silu(of._lin(sd, f"{prefix}time_mlp.linear_1", temb)))
    T = x.shape[-1]
    h = torch.cat([x, mu, spks.unsqueeze(-1).expand(-1, -1, T), cond], dim=1)
    g = cfg.est_groups

    def stage(name, h):
        h = resnet_block(sd, f"{name}.0", h, temb, g).transpose(1, 2)
        ones = torch.ones(h.shape[0], h.shape[1], h.shape[1])   # the 0/1 mask product ADDED to the scores (decoder.py:258)
        for j in range(cfg.est_n_blocks):
            h = of.transformer_block(sd, f"{name}.1.{j}", h, ones, cfg.est_heads, cfg.est_head_dim)
        return h.transpose(1, 2)

    h = stage(f"{prefix}down_blocks.0", h)
    skip0 = h
    h = F.conv1d(h, sd[f"{prefix}down_blocks.0.2.conv.weight"], sd[f"{prefix}down_blocks.0.2.conv.bias"], stride=2, padding=1)
    h = stage(f"{prefix}down_blocks.1", h)
    skip1 = h
    h = F.conv1d(h, sd[f"{prefix}down_blocks.1.2.weight"], sd[f"{prefix}down_blocks.1.2.bias"], padding=1)
    for i in range(cfg.est_mid_blocks):
        h = stage(f"{prefix}mid_blocks.{i}", h)
    h = stage(f"{prefix}up_blocks.0", torch.cat([h[:, :, :skip1.shape[-1]], skip1], dim=1))
    h = F.conv_transpose1d(h, sd[f"{prefix}up_blocks.0.2.conv.weight"], sd[f"{prefix}up_blocks.0.2.conv.bias"], stride=2, padding=1)
    h = stage(f"{prefix}up_blocks.1", torch.cat([h[:, :, :skip0.shape[-1]], skip0], dim=1))
    h = F.conv1d(h, sd[f"{prefix}up_blocks.1.2.weight"], sd[f"{prefix}up_blocks.1.2.bias"], padding=1)
    h = block1d(sd, f"{prefix}final_block", h, g)
    return F.conv1d(h, sd[f"{prefix}final_proj.weight"], sd[f"{prefix}final_proj.bias"])


def regulator_inference(sd, cfg, x1, x2, mel_len1, mel_len2, sample_rate):
    """InterpolateRegulator.inference, length_regulator.py:49-70: prompt / head / middle / tail interpolated separately."""
    n20 = int(20 / cfg.input_frame_rate * sample_rate / cfg.hop_size)
    it = lambda v, size: F.interpolate(v.transpose(1, 2).contiguous(), size=size, mode="linear")
    if x2.shape[1] > 40:
        x2 = torch.cat([it(x2[:, :20], n20), it(x2[:, 20:-20], mel_len2 - 2 * n20), it(x2[:, -20:], n20)], dim=2)
    else:
        x2 = it(x2, mel_len2)
    x = torch.cat([it(x1, mel_len1), x2], dim=2) if x1.shape[1] != 0 else x2
    p = "length_regulator.model"
    for i in range(cfg.reg_layers):
        x = F.conv1d(x, sd[f"{p}.{3 * i}.weight"], sd[f"{p}.{3 * i}.bias"], padding=1)
        x = F.mish(_gn(sd, f"{p}.{3 * i + 1}", x, cfg.reg_groups))
    n = 3 * cfg.reg_layers
    return F.conv1d(x, sd[f"{p}.{n}.weight"], sd[f"{p}.{n}.bias"]).transpose(1, 2)


def mel_len_of(cfg, n_tokens, sample_rate):
    return int(n_tokens / cfg.input_frame_rate * sample_rate / cfg.hop_size)    # flow.py:143


def inference(sd, cfg, token, prompt_token, prompt_feat, embedding, flow_cache, sample_rate, z, n_timesteps=10,
              return_mu=False):
    """MaskedDiffWithXvec.inference.  token (1,Ng), prompt_token (1,Np), prompt_feat (1,T1,80), embedding (1,D),
    flow_cache (1,80,Tc,2), z (1,80,T1+T2) = the torch.randn_like draw of flow_matching.py:56 (injected).
    -> (mel (1,80,T2), new flow_cache (1,80,T1+34,2))."""
    assert token.shape[0] == 1
    spks = of._lin(sd, "spk_embed_affine_layer", F.normalize(embedding, dim=1))
    n1, n2 = prompt_token.shape[1], token.shape[1]
    tok = torch.cat([prompt_token, token], dim=1)
    emb = F.embedding(torch.clamp(tok, min=0).long(), sd["input_embedding.weight"])
    T = tok.shape[1]
    xs, pos_emb = of.embed(sd, "encoder.embed", emb, cfg.enc_dim)
    masks = torch.ones(1, 1, T, dtype=torch.bool)
    for i in range(cfg.enc_blocks):
        xs = of.conformer_layer(sd, f"encoder.encoders.{i}", xs, masks, pos_emb, cfg.enc_heads)
    h = of._lin(sd, "encoder_proj", of._ln(sd, "encoder.after_norm", xs, 1e-5))
    mel_len1, mel_len2 = prompt_feat.shape[1], mel_len_of(cfg, n2, sample_rate)
    h = regulator_inference(sd, cfg, h[:, :n1], h[:, n1:], mel_len1, mel_len2, sample_rate)
    Tm = mel_len1 + mel_len2
    cond = torch.zeros(1, Tm, cfg.output_size)
    cond[:, :mel_len1] = prompt_feat
    cond = cond.transpose(1, 2)
    mu = h.transpose(1, 2).contiguous()
    # ConditionalCFM.forward, flow_matching.py:56-66
    z = z.clone()
    cache_size = min(flow_cache.shape[2], Tm)
    if cache_size != 0:
        z[:, :, :cache_size] = flow_cache[:, :, :cache_size, 0]
        mu[:, :, :cache_size] = flow_cache[:, :, :cache_size, 1]
    new_cache = torch.stack([torch.cat([z[:, :, :mel_len1], z[:, :, -34:]], dim=2),
                             torch.cat([mu[:, :, :mel_len1], mu[:, :, -34:]], dim=2)], dim=-1)
    if return_mu:
        return mu, new_cache
    est = lambda x, mask, mu_, t, sp, c: estimator_forward(sd, cfg, x, mu_, t, sp, c)
    mask = torch.ones(1, 1, Tm)
    feat = of.solve_euler(sd, cfg, z, of.t_span_cosine(n_timesteps), mu, mask, spks, cond, estimator=est)
    return feat[:, :, mel_len1:], new_cache
