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
