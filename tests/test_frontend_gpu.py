"""GPU: prompt-feature front half (mel_spectrogram as two cv_gemm calls + cv_stft_magnitude + cv_log_clamp_channels_first)
against the reference-minted golden and the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_mel_vs_reference_golden(golden_dir):
    from cosyvoice_amd.frontend import MelSpectrogram, extract_speech_feat
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, "frontend_mel.npz")).items()}
    fe = MelSpectrogram()
    mel = fe(g["y"].cuda()).cpu()
    assert mel.shape == g["mel"].shape and mel.dtype == torch.float32
    # log domain: 1e-3 absolute = 0.1 % of the linear mel energy
    assert (mel - g["mel"]).abs().max().item() < 1e-3
    feat, n = extract_speech_feat(fe, g["y"][:1].cuda())
    assert feat.shape == (1, g["mel"].shape[2], 80) and int(n) == g["mel"].shape[2]


@pytest.mark.parametrize("B,S", [(1, 24000), (3, 4801), (1, 1921), (2, 721), (1, 240000)])
def test_mel_vs_oracle_lengths(B, S):
    from cosyvoice_amd.frontend import MelSpectrogram, slaney_mel_basis
    from oracle import frontend as ofe
    torch.manual_seed(S)
    y = (torch.randn(B, S) * 0.3).clamp(-1, 1)
    ref = ofe.mel_spectrogram(y, torch.from_numpy(slaney_mel_basis(24000, 1920, 80, 0, 8000)))
    mel = MelSpectrogram()(y.cuda()).cpu()
    assert mel.shape == ref.shape == (B, 80, (S + 1440 - 1920) // 480 + 1)
    assert (mel - ref).abs().max().item() < 1e-3
