"""BigVGAN anti-aliased activation on MI355X: drop-in for the reference's fused ``Activation1d``
(/root/reference/cosyvoice/BigVGAN/alias_free_activation/cuda/activation1d.py:36-76), whose CUDA extension
(anti_alias_activation_cuda.cu) is the reference's only native kernel.  Same call contract: ``forward(x [B,C,T])`` with
a Snake / SnakeBeta activation object carrying ``alpha`` (and ``beta``) and ``alpha_logscale``."""
import ctypes as C
import math

import torch

from . import _lib as L


def kaiser_sinc_filter12(cutoff: float = 0.25, half_width: float = 0.3) -> torch.Tensor:
    """12-tap kaiser-windowed sinc of UpSample1d/DownSample1d(ratio 2) — alias_free_activation/torch/filter.py:62-94."""
    k, half = 12, 6
    A = 2.285 * (half - 1) * math.pi * 4 * half_width + 7.95
    beta = 0.1102 * (A - 8.7) if A > 50.0 else (0.5842 * (A - 21) ** 0.4 + 0.07886 * (A - 21.0) if A >= 21.0 else 0.0)
    window = torch.kaiser_window(k, beta=beta, periodic=False)
    time = torch.arange(-half, half) + 0.5
    f = 2 * cutoff * window * torch.sinc(2 * cutoff * time)
    return (f / f.sum()).to(torch.float32)


class Activation1d:
    def __init__(self, activation, up_ratio: int = 2, down_ratio: int = 2, up_kernel_size: int = 12, down_kernel_size: int = 12,
                 device: str = "cuda"):
        if (up_ratio, down_ratio, up_kernel_size, down_kernel_size) != (2, 2, 12, 12):
            raise ValueError("the fused kernel is hard-wired to ratio 2 / 12 taps, as the reference's (activation1d.py:16-19)")
        self.act = activation
        self.device = torch.device(device)
        f = kaiser_sinc_filter12().to(self.device)
        self.up_filter, self.down_filter = f.contiguous(), f.clone().contiguous()

    def _log_params(self):
        alpha = self.act.alpha.detach().to(self.device, torch.float32)
        beta = self.act.beta.detach().to(self.device, torch.float32) if hasattr(self.act, "beta") else alpha  # Snake: beta = alpha
        if not getattr(self.act, "alpha_logscale", False):  # exp is baked into the kernel (activation1d.py:66-71)
            alpha, beta = torch.log(alpha), torch.log(beta)
        return alpha.contiguous(), beta.contiguous()

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("cosyvoice_amd.bigvgan needs device tensors (no CPU path)")
        x = x.contiguous()
        B, Cc, T = x.shape
        y = torch.empty_like(x)
        a, b = self._log_params()
        L.check(L.lib().cv_anti_alias_act(C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), L.TORCH_DT[x.dtype], B, Cc, T,
                                          C.c_void_p(self.up_filter.data_ptr()), C.c_void_p(self.down_filter.data_ptr()),
                                          C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), L.stream_ptr()), "cv_anti_alias_act")
        return y

    __call__ = forward
