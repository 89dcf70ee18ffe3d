"""GPU: the HIP HiFT path (cosyvoice_amd.hift, through the C ABI) against the CPU oracle and the reference-minted
golden fixtures.  Tolerances: waveform L-inf <= 2e-3 at +-0.99 full scale (fp32 path: 1e-4)."""
import os

import numpy as np
import pytest
import torch

from cosyvoice_amd.config import HiftConfig
from cosyvoice_amd.weights import hift_state_dict

pytestmark = pytest.mark.gpu

CFGS = {"tiny": HiftConfig.tiny(), "v2": HiftConfig.v2(), "v1": HiftConfig.v1()}


def _golden(golden_dir, tag):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, f"hift_{tag}.npz")).items()}


@pytest.mark.parametrize("tag", ["tiny", "v2", "v1"])
@pytest.mark.parametrize("dt,tol", [(torch.float32, 1e-4), (torch.bfloat16, 2e-3), (torch.float16, 2e-3)])
def test_decode_vs_golden(golden_dir, tag, dt, tol):
    from cosyvoice_amd.hift import HiFTGenerator
    cfg = CFGS[tag]
    g = _golden(golden_dir, tag)
    m = HiFTGenerator(cfg, dtype=dt).load_state_dict(hift_state_dict(cfg))
    wav = m.decode(g["mel"].cuda(), g["s"].cuda()).cpu()
    assert wav.shape == g["wav"].shape
    err = (wav - g["wav"]).abs().max().item()
    assert err < tol, err


@pytest.mark.parametrize("tag", ["tiny", "v2"])
def test_f0_and_source_vs_golden(golden_dir, tag):
    from cosyvoice_amd.hift import HiFTGenerator
    cfg = CFGS[tag]
    g = _golden(golden_dir, tag)
    m = HiFTGenerator(cfg, dtype=torch.float32).load_state_dict(hift_state_dict(cfg))
    f0 = m.f0_predictor(g["mel"].cuda())
    assert (f0.cpu() - g["f0"]).abs().max().item() < 2e-3  # Hz, values up to ~130
    # source with the reference's own random draws; the reference scan is an order-dependent fp32 cumsum (H4):
    # the HIP kernel computes the exact (fp64) phase, tolerance covers the reference's own rounding
    s = m.source(g["f0"].cuda(), g["phase_vec"].cuda(), g["noise"].cuda())
    assert (s.cpu() - g["src"].reshape(s.shape)).abs().max().item() < 5e-3


def test_inference_vs_oracle_full_shape():
    """Full BASELINE shape (v2, 80x500 mel -> 240 000 samples) against the oracle with injected randoms."""
    from cosyvoice_amd.hift import HiFTGenerator
    from oracle import hift as oh
    cfg = HiftConfig.v2()
    sd = hift_state_dict(cfg)
    torch.manual_seed(0)
    T = 100
    mel = torch.clamp(torch.randn(1, 80, T) * 2 - 6, -11.5, 2.0)
    ph, nz = oh.draw_source_randoms(cfg, 1, T * cfg.total_upsample, seed=5)
    wav_ref, s_ref = oh.inference(sd, cfg, mel, None, ph, nz, scan_dtype=torch.float64)
    m = HiFTGenerator(cfg, dtype=torch.float32).load_state_dict(sd)
    wav, s = m.inference(mel.cuda(), torch.zeros(1, 1, 0), ph.cuda(), nz.cuda())
    assert s.shape == s_ref.shape and wav.shape == wav_ref.shape
    assert (s.cpu() - s_ref).abs().max().item() < 1e-3
    assert (wav.cpu() - wav_ref).abs().max().item() < 1e-3
    # cache_source overwrite path (generator.py:408-409)
    cache = s_ref[:, :, :960].clone() * 0.5
    wav2, s2 = m.inference(mel.cuda(), cache.cuda(), ph.cuda(), nz.cuda())
    assert torch.allclose(s2[:, :, :960].cpu(), cache, atol=1e-6)
    wav_ref2, _ = oh.inference(sd, cfg, mel, cache, ph, nz, scan_dtype=torch.float64)
    assert (wav2.cpu() - wav_ref2).abs().max().item() < 1e-3


def test_batched_decode_matches_single():
    from cosyvoice_amd.hift import HiFTGenerator
    cfg = HiftConfig.tiny()
    sd = hift_state_dict(cfg)
    torch.manual_seed(1)
    mel = torch.clamp(torch.randn(3, 80, 20) * 2 - 6, -11.5, 2.0).cuda()
    s = (torch.randn(3, 1, 20 * cfg.total_upsample) * 0.05).cuda()
    m = HiFTGenerator(cfg, dtype=torch.float32).load_state_dict(sd)
    wb = m.decode(mel, s).clone()
    for b in range(3):
        w1 = m.decode(mel[b:b + 1], s[b:b + 1])
        assert (w1[0] - wb[b]).abs().max().item() < 1e-6


@pytest.mark.parametrize("tag", ["tiny", "v2", "v1"])
@pytest.mark.parametrize("dt,mode", [(torch.float32, "exact"), (torch.float32, "bf16x3"), (torch.float16, "exact")])
def test_stage_abi_decode_equals_python_composed_decode(tag, dt, mode):
    """cv_hift_decode_enqueue (the library composes the ~100 launches of HiFTGenerator.decode from a descriptor) vs the launch
    sequence cosyvoice_amd/hift.py issues itself: same kernels, same order -> bit-identical waveform."""
    from cosyvoice_amd.hift import HiFTGenerator
    cfg = CFGS[tag]
    m = HiFTGenerator(cfg, dtype=dt, f32_products=mode).load_state_dict(hift_state_dict(cfg))
    assert m.use_stage_abi
    torch.manual_seed(3)
    B, T = 2, 23
    mel = torch.clamp(torch.randn(B, 80, T) * 2 - 6, -11.5, 2.0).cuda()
    s = (torch.randn(B, 1, T * cfg.total_upsample) * 0.05).cuda()
    w_abi = m.decode(mel, s).clone()
    m.use_stage_abi = False
    w_py = m.decode(mel, s).clone()
    torch.cuda.synchronize()
    assert torch.isfinite(w_abi).all() and w_abi.abs().max().item() > 0 and torch.equal(w_abi, w_py)


@pytest.mark.parametrize("tag", ["tiny", "v2"])
def test_presplit_storage_is_bit_identical(tag, monkeypatch):
    """bf16x3 decoder convs with PRE-SPLIT activation / weight storage (hi / lo bf16 planes per 4 values, written once by the producing
    epilogue) against the plain bf16x3 launches that split every operand tile as it is staged: same hi / lo bits, same MFMAs — bit-identical
    on the implicit-GEMM kernel (the default).  The opt-in activation-window kernel (CV_CONV_WIN=1) sums chunk-major (64 channels at a time,
    then taps) instead of tap-major: same products, fp32 re-association only."""
    from cosyvoice_amd.hift import HiFTGenerator
    cfg = CFGS[tag]
    sd = hift_state_dict(cfg)
    torch.manual_seed(5)
    B, T = 2, 31
    mel = torch.clamp(torch.randn(B, 80, T) * 2 - 6, -11.5, 2.0).cuda()
    s = (torch.randn(B, 1, T * cfg.total_upsample) * 0.05).cuda()
    outs = {}
    for flag, win in (("1", "0"), ("0", "0"), ("1", "1")):
        monkeypatch.setenv("CV_HIFT_PRESPLIT", flag)
        monkeypatch.setenv("CV_CONV_WIN", win)
        m = HiFTGenerator(cfg, dtype=torch.float32, f32_products="bf16x3").load_state_dict(sd)
        assert m.presplit == (flag == "1")
        outs[flag + win] = m.decode(mel, s).clone()
    torch.cuda.synchronize()
    assert outs["10"].abs().max().item() > 0 and torch.equal(outs["10"], outs["00"])
    assert (outs["11"] - outs["10"]).abs().max().item() < 2e-6
