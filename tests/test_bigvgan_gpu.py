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


@pytest.mark.parametrize("idt,odt,tol", [(torch.float32, torch.float32, 3e-5), (torch.float32, torch.bfloat16, 2e-2),
                                         (torch.float16, torch.float16, 8e-3), (torch.bfloat16, torch.float32, 6e-2)])
@pytest.mark.parametrize("B,T,C,ld", [(2, 1000, 192, 192), (1, 33, 96, 104), (1, 1, 3, 8), (2, 257, 70, 72)])
def test_channels_last_form_vs_oracle(idt, odt, tol, B, T, C, ld):
    """cv_anti_alias_act_cl (the AMP blocks' layout) against the same oracle, incl. padded channel strides."""
    from cosyvoice_amd.bigvgan import anti_alias_act_cl, kaiser_sinc_filter12
    from oracle import bigvgan as ob
    torch.manual_seed(1)
    x = (torch.randn(B, C, T) * 2).to(idt)
    a, b = torch.randn(C) * 0.5, torch.randn(C) * 0.5
    ref = ob.anti_alias_activation(x.float(), a, b).transpose(1, 2)
    xcl = torch.zeros(B, T, ld, dtype=idt, device="cuda")
    xcl[:, :, :C] = x.transpose(1, 2).cuda()
    y = torch.full((B, T, ld), 7.0, dtype=odt, device="cuda")
    f = kaiser_sinc_filter12().cuda()
    anti_alias_act_cl(xcl, y, C, f, f.clone(), a.cuda(), b.cuda())
    torch.cuda.synchronize()
    assert (y[:, :, :C].float().cpu() - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item())
    assert (y[:, :, C:].float() == 7.0).all()   # padded channels untouched


def test_generator_vs_reference_golden(golden_dir):
    """The full BigVGAN generator (token embedding -> encoder_proj -> conv_pre + speaker conditioning -> transposed-conv
    stages with AMP blocks -> tanh) against the reference's own BigVGAN.forward output."""
    from cosyvoice_amd.bigvgan import BigVGAN
    from cosyvoice_amd.config import BigVGANConfig
    from cosyvoice_amd.weights import bigvgan_state_dict
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, "bigvgan_tiny.npz")).items()}
    cfg = BigVGANConfig.tiny()
    m = BigVGAN(cfg, dtype=torch.float32).load_state_dict(bigvgan_state_dict(cfg, seed=int(g["seed"])))
    wav, (mel, _) = m(dict(speech_token=g["token"], speech_token_len=g["token_len"], embedding=g["embedding"]), "cuda")
    assert wav.shape == g["wav"].shape and mel.shape == g["mel"].shape
    assert (wav.cpu() - g["wav"]).abs().max().item() < 2e-4
    assert (mel.cpu() - g["mel"]).abs().max().item() < 2e-4


@pytest.mark.parametrize("dt,tol", [(torch.float32, 3e-4), (torch.float16, 3e-2), (torch.bfloat16, 2e-1)])
def test_generator_vs_oracle_ragged_batch(dt, tol):
    from cosyvoice_amd.bigvgan import BigVGAN
    from cosyvoice_amd.config import BigVGANConfig
    from cosyvoice_amd.weights import bigvgan_state_dict
    from oracle import bigvgan as ob
    cfg = BigVGANConfig.tiny()
    sd = bigvgan_state_dict(cfg, seed=9)
    g = torch.Generator().manual_seed(3)
    B, N = 3, 37
    token = torch.randint(-1, cfg.vocab_size, (B, N), generator=g)   # -1: the reference clamps negative ids to 0
    token_len = torch.tensor([37, 20, 1])
    emb = torch.randn(B, cfg.speaker_embedding_dim, generator=g)
    ref_wav, ref_mel = ob.bigvgan_forward(sd, cfg, token, token_len, emb)
    m = BigVGAN(cfg, dtype=dt).load_state_dict(sd)
    wav, (mel, _) = m(dict(speech_token=token, speech_token_len=token_len, embedding=emb), "cuda")
    assert wav.shape == (B, N * cfg.total_upsample)
    assert (wav.cpu() - ref_wav).abs().max().item() < tol
    assert (mel.cpu() - ref_mel).abs().max().item() < tol * max(1.0, ref_mel.abs().max().item())
    # second call reuses the workspaces
    wav2, _ = m(dict(speech_token=token, speech_token_len=token_len, embedding=emb), "cuda")
    assert torch.equal(wav, wav2)


def test_generator_full_config_vs_oracle():
    """Reference-default architecture (1536 channels, x1024 upsampling, 18 AMP blocks): fp32-MFMA generator vs the oracle."""
    from cosyvoice_amd.bigvgan import BigVGAN
    from cosyvoice_amd.config import BigVGANConfig
    from cosyvoice_amd.weights import bigvgan_state_dict
    from oracle import bigvgan as ob
    cfg = BigVGANConfig.full()
    sd = bigvgan_state_dict(cfg, seed=2)
    g = torch.Generator().manual_seed(5)
    B, N = 1, 10
    token = torch.randint(0, cfg.vocab_size, (B, N), generator=g)
    token_len = torch.tensor([N])
    emb = torch.randn(B, cfg.speaker_embedding_dim, generator=g)
    torch.set_num_threads(16)
    ref_wav, ref_mel = ob.bigvgan_forward(sd, cfg, token, token_len, emb)
    m = BigVGAN(cfg, dtype=torch.float32).load_state_dict(sd)
    wav, (mel, _) = m(dict(speech_token=token, speech_token_len=token_len, embedding=emb), "cuda")
    assert wav.shape == (B, N * 1024) and torch.isfinite(wav).all()
    assert (wav.cpu() - ref_wav).abs().max().item() < 2e-3
    assert (mel.cpu() - ref_mel).abs().max().item() < 1e-3 * max(1.0, ref_mel.abs().max().item())


def test_generator_with_injected_encoders():
    """encoder1 / encoder2 are injected modules in the reference (bigvgan.py:273-285,395-402): each doubles the frame rate;
    with encoder2 present mel_proj reads the encoder output.  Stand-in encoders (nearest x2 + a fixed mixing) on both sides."""
    from cosyvoice_amd.bigvgan import BigVGAN
    from cosyvoice_amd.config import BigVGANConfig
    from cosyvoice_amd.weights import bigvgan_state_dict
    from oracle import bigvgan as ob
    cfg = BigVGANConfig.tiny()
    sd = bigvgan_state_dict(cfg, seed=4)
    D = cfg.input_size
    # with encoder2 the reference sizes encoder_proj / mel_proj by encoder2.output_size(): same width here, new mel_proj
    g = torch.Generator().manual_seed(8)
    sd["mel_proj.weight"] = torch.randn(cfg.mel_bin, D, generator=g) / D ** 0.5
    mix = torch.randn(D, D, generator=g) / D ** 0.5

    def enc(x, n):   # (B,T,D) -> (B,2T,D)
        return torch.tanh(x.repeat_interleave(2, dim=1) @ mix.to(x.device)), None

    B, N = 2, 9
    token = torch.randint(0, cfg.vocab_size, (B, N), generator=g)
    token_len = torch.tensor([9, 5])
    emb = torch.randn(B, cfg.speaker_embedding_dim, generator=g)
    ref_wav, ref_mel = ob.bigvgan_forward(sd, cfg, token, token_len, emb, encoder1=enc, encoder2=enc)
    m = BigVGAN(cfg, dtype=torch.float32, encoder1=enc, encoder2=enc).load_state_dict(sd)
    wav, (mel, _) = m(dict(speech_token=token, speech_token_len=token_len, embedding=emb), "cuda")
    assert wav.shape == ref_wav.shape == (B, 4 * N * cfg.total_upsample) and mel.shape == ref_mel.shape == (B, 4 * N, cfg.mel_bin)
    assert (wav.cpu() - ref_wav).abs().max().item() < 3e-4
    assert (mel.cpu() - ref_mel).abs().max().item() < 3e-4
