"""GPU: cv_anti_alias_act (the HIP counterpart of the reference's only CUDA kernel) vs the golden / oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class _SnakeBeta:  # duck-typed like BigVGAN/nnet/activations.py SnakeBeta (alpha, beta, alpha_logscale)
    def __init__(self, alpha, beta, logscale):
        self.alpha, self.beta, self.alpha_logscale = alpha, beta, logscale


def test_vs_reference_golden(golden_dir):
    from cosyvoice_amd.bigvgan import Activation1d
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, "bigvgan_act.npz")).items()}
    m = Activation1d(_SnakeBeta(g["alpha_log"], g["beta_log"], True))
    y = m(g["x"].cuda()).cpu()
    assert (y - g["y"]).abs().max().item() < 2e-5
    # linear-scale parameters take the log path of activation1d.py:66-71
    m2 = Activation1d(_SnakeBeta(torch.exp(g["alpha_log"]), torch.exp(g["beta_log"]), False))
    assert (m2(g["x"].cuda()).cpu() - g["y"]).abs().max().item() < 2e-5


@pytest.mark.parametrize("dt,tol", [(torch.float32, 3e-5), (torch.bfloat16, 6e-2), (torch.float16, 8e-3)])
@pytest.mark.parametrize("B,C,T", [(1, 96, 24000), (2, 192, 1000), (1, 3, 1), (1, 2, 255), (1, 2, 257)])
def test_vs_oracle_shapes(dt, tol, B, C, T):
    from cosyvoice_amd.bigvgan import Activation1d
    from oracle import bigvgan as ob
    torch.manual_seed(0)
    x = (torch.randn(B, C, T) * 2).to(dt)
    a, b = torch.randn(C) * 0.5, torch.randn(C) * 0.5
    ref = ob.anti_alias_activation(x.float(), a, b)
    y = Activation1d(_SnakeBeta(a, b, True))(x.cuda()).float().cpu()
    assert y.shape == ref.shape
    assert (y - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item())
