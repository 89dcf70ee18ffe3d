"""Oracle for BigVGAN's anti-aliased activation (TEST INFRASTRUCTURE — see oracle/__init__.py).

Restates /root/reference/cosyvoice/BigVGAN/alias_free_activation/torch/{act.py:26-31, resample.py:10-58,
filter.py:62-133} and SnakeBeta (BigVGAN/nnet/activations.py:109-122, alpha_logscale=True) — the torch path whose
result the reference's CUDA kernel (cuda/anti_alias_activation_cuda.cu) is documented to equal."""
import math

import torch
import torch.nn.functional as F


def kaiser_sinc_filter1d(cutoff: float, half_width: float, kernel_size: int) -> torch.Tensor:
    # filter.py:62-94
    even = kernel_size % 2 == 0
    half_size = kernel_size // 2
    delta_f = 4 * half_width
    A = 2.285 * (half_size - 1) * math.pi * delta_f + 7.95
    if A > 50.0:
        beta = 0.1102 * (A - 8.7)
    elif A >= 21.0:
        beta = 0.5842 * (A - 21) ** 0.4 + 0.07886 * (A - 21.0)
    else:
        beta = 0.0
    window = torch.kaiser_window(kernel_size, beta=beta, periodic=False)
    time = (torch.arange(-half_size, half_size) + 0.5) if even else (torch.arange(kernel_size) - half_size)
    filt = 2 * cutoff * window * torch.sinc(2 * cutoff * time)
    return (filt / filt.sum()).view(1, 1, kernel_size)


def upsample2(x: torch.Tensor, filt: torch.Tensor) -> torch.Tensor:
    # UpSample1d(ratio=2, kernel 12), resample.py:10-36
    C = x.shape[1]
    ratio, k = 2, filt.shape[-1]
    pad = k // ratio - 1
    pad_left = pad * ratio + (k - ratio) // 2
    pad_right = pad * ratio + (k - ratio + 1) // 2
    x = F.pad(x, (pad, pad), mode="replicate")
    x = ratio * F.conv_transpose1d(x, filt.expand(C, -1, -1), stride=ratio, groups=C)
    return x[..., pad_left:-pad_right]


def downsample2(x: torch.Tensor, filt: torch.Tensor) -> torch.Tensor:
    # DownSample1d -> LowPassFilter1d(stride 2, replicate pad 5/6), filter.py:121-133
    C = x.shape[1]
    k = filt.shape[-1]
    x = F.pad(x, (k // 2 - 1, k // 2), mode="replicate")
    return F.conv1d(x, filt.expand(C, -1, -1), stride=2, groups=C)


def snakebeta_log(x: torch.Tensor, alpha_log: torch.Tensor, beta_log: torch.Tensor) -> torch.Tensor:
    # activations.py:109-122 with alpha_logscale=True
    a = torch.exp(alpha_log)[None, :, None]
    b = torch.exp(beta_log)[None, :, None]
    return x + (1.0 / (b + 1e-9)) * torch.sin(x * a) ** 2


def anti_alias_activation(x: torch.Tensor, alpha_log: torch.Tensor, beta_log: torch.Tensor) -> torch.Tensor:
    """Activation1d.forward (act.py:26-31): x [B,C,T] -> [B,C,T]."""
    f = kaiser_sinc_filter1d(0.25, 0.3, 12)
    return downsample2(snakebeta_log(upsample2(x, f), alpha_log, beta_log), f)


# ------------------------------------------------------------------ full generator (BigVGAN/bigvgan.py:257-438)
def _wn(sd, name):
    from cosyvoice_amd.weights import fold_weight_norm
    return fold_weight_norm(sd, name)


def amp_block1(sd, name: str, x: torch.Tensor, k: int, dils) -> torch.Tensor:
    """AMPBlock1.forward (bigvgan.py:128-137): per dilation  x = conv2(act2(conv1(act1(x)))) + x."""
    for j, d in enumerate(dils):
        a1, a2 = 2 * j, 2 * j + 1
        xt = anti_alias_activation(x, sd[f"{name}.activations.{a1}.act.alpha"], sd[f"{name}.activations.{a1}.act.beta"])
        xt = F.conv1d(xt, _wn(sd, f"{name}.convs1.{j}"), sd[f"{name}.convs1.{j}.bias"], dilation=d, padding=(k * d - d) // 2)
        xt = anti_alias_activation(xt, sd[f"{name}.activations.{a2}.act.alpha"], sd[f"{name}.activations.{a2}.act.beta"])
        xt = F.conv1d(xt, _wn(sd, f"{name}.convs2.{j}"), sd[f"{name}.convs2.{j}.bias"], padding=(k - 1) // 2)
        x = xt + x
    return x


def bigvgan_forward(sd, cfg, token: torch.Tensor, token_len: torch.Tensor, embedding: torch.Tensor, encoder1=None, encoder2=None):
    """BigVGAN.forward (bigvgan.py:384-438).  token (B,N) int, token_len (B,), embedding (B, D_spk) ->
    (wav (B, N' * prod(upsample_rates)), mel_feat_out (B, N', mel_bin)); encoder1 / encoder2: the injected x2-upsampling
    encoders ``(x, x_len) -> (y, mask)`` (:395-402), None = tokens go straight to encoder_proj (N' = N)."""
    B, N = token.shape
    spk = embedding.unsqueeze(-1).float()
    mask = (torch.arange(N)[None, :] < token_len[:, None]).float().unsqueeze(-1)          # ~make_pad_mask
    x = F.embedding(torch.clamp(token, min=0).long(), sd["input_embedding.weight"]) * mask
    mel = None
    if encoder1 is not None:
        x, _ = encoder1(x, token_len)
        token_len = token_len * 2
    if encoder2 is not None:
        x, _ = encoder2(x, token_len)
        token_len = token_len * 2
        mel = F.linear(x, sd["mel_proj.weight"], sd["mel_proj.bias"])
    x = F.linear(x, sd["encoder_proj.weight"], sd["encoder_proj.bias"]).transpose(1, 2)
    x = F.conv1d(x, _wn(sd, "conv_pre"), sd["conv_pre.bias"], padding=3)
    x = x + F.conv1d(spk, sd["cond_layer.weight"], sd["cond_layer.bias"])
    if mel is None:
        mel = F.linear(x.transpose(1, 2), sd["mel_proj.weight"], sd["mel_proj.bias"])
    nk = len(cfg.resblock_kernel_sizes)
    for i, (u, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
        x = F.conv_transpose1d(x, _wn(sd, f"ups.{i}.0"), sd[f"ups.{i}.0.bias"], stride=u, padding=(k - u) // 2)
        if cfg.cond_in_each_up_layer:
            x = x + F.conv1d(spk, sd[f"conds.{i}.weight"], sd[f"conds.{i}.bias"])
        xs = None
        for j, (k2, dils) in enumerate(zip(cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes)):
            y = amp_block1(sd, f"resblocks.{i * nk + j}", x, k2, dils)
            xs = y if xs is None else xs + y
        x = xs / nk
    x = anti_alias_activation(x, sd["activation_post.act.alpha"], sd["activation_post.act.beta"])
    x = F.conv1d(x, _wn(sd, "conv_post"), sd["conv_post.bias"], padding=3)
    return torch.tanh(x).squeeze(1), mel
