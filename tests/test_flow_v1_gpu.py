"""GPU: CosyVoice-v1 flow MaskedDiffWithXvec against the reference-minted golden
(tests/golden/flow_v1_tiny.npz: the reference's own ConformerEncoder / InterpolateRegulator / ConditionalCFM / non-causal
two-level estimator) and the new GroupNorm / linear-interpolation kernels against torch."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from cosyvoice_amd.config import FlowV1Config
from cosyvoice_amd.weights import flow_v1_state_dict

pytestmark = pytest.mark.gpu


def _golden(golden_dir):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, "flow_v1_tiny.npz")).items()}


@pytest.mark.parametrize("B,T,C,G", [(2, 37, 256, 8), (1, 106, 80, 1), (2, 500, 256, 8), (3, 1, 64, 4)])
def test_groupnorm_cl_vs_torch(B, T, C, G):
    from cosyvoice_amd import ops
    g = torch.Generator().manual_seed(T)
    x = (torch.randn(B, T, C, generator=g) * 2 + 0.7)
    gamma, beta, add = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g), torch.randn(C, generator=g)
    ref = F.mish(F.group_norm(x.transpose(1, 2), G, gamma, beta, 1e-5)).transpose(1, 2) + add
    xd = x.cuda()
    part = ops.groupnorm_workspace(B, T, G, "cuda")
    o32 = torch.empty(B, T, C, device="cuda")
    o16 = torch.empty(B, T, 2 * C, device="cuda", dtype=torch.float16)[:, :, C:]      # strided 16-bit output
    ops.groupnorm_cl(xd, G, gamma.cuda(), beta.cuda(), 1e-5, part, act=ops.ACT_MISH, add=add.cuda(), out_f32=o32, out_act=o16)
    assert (o32.cpu() - ref).abs().max().item() < 2e-5
    assert (o16.float().cpu() - ref).abs().max().item() < 6e-3
    ops.groupnorm_cl(xd, G, gamma.cuda(), beta.cuda(), 1e-5, part, out_f32=xd)        # in place, no activation
    assert (xd.cpu() - F.group_norm(x.transpose(1, 2), G, gamma, beta, 1e-5).transpose(1, 2)).abs().max().item() < 2e-5


@pytest.mark.parametrize("Tin,Tout", [(20, 34), (12, 20), (10, 18), (50, 86), (7, 3), (1, 5)])
def test_interp_linear_cl_vs_torch(Tin, Tout):
    from cosyvoice_amd import ops
    x = torch.randn(Tin, 80, generator=torch.Generator().manual_seed(Tin))
    ref = F.interpolate(x.t().unsqueeze(0), size=Tout, mode="linear")[0].t()
    y = torch.empty(Tout, 80, device="cuda")
    ops.interp_linear_cl(x.cuda(), y)
    assert (y.cpu() - ref).abs().max().item() < 1e-6


@pytest.mark.parametrize("dt,tol", [(torch.float16, 2e-2), (torch.bfloat16, 1.5e-1)])
def test_estimator_v1_vs_reference_golden(golden_dir, dt, tol):
    from cosyvoice_amd.flow_v1 import MaskedDiffWithXvec
    g = _golden(golden_dir)
    c = FlowV1Config.tiny()
    m = MaskedDiffWithXvec(c, dtype=dt).load_state_dict(flow_v1_state_dict(c))
    mask = torch.ones(2, 1, g["est_x"].shape[2])
    out = m.decoder.estimator(g["est_x"], mask, g["est_mu"], g["est_t"], g["est_spks"], g["est_cond"]).cpu()   # odd T = 37
    err = (out - g["est_out"]).abs()
    print(f"v1 estimator[{dt}] Linf {err.max():.3e} L1 {err.mean():.3e} (|ref| max {g['est_out'].abs().max():.2f})")
    assert err.max().item() < tol * max(1.0, g["est_out"].abs().max().item())


@pytest.mark.parametrize("dt,l1,linf", [(torch.float16, 2e-3, 2e-2), (torch.bfloat16, 1.5e-2, 1.5e-1)])
def test_inference_vs_reference_golden(golden_dir, dt, l1, linf):
    """Two chunks, the second inheriting the first one's flow cache; z injected = the reference's seeded torch.randn_like draw."""
    from cosyvoice_amd.flow_v1 import MaskedDiffWithXvec
    g = _golden(golden_dir)
    c = FlowV1Config.tiny()
    m = MaskedDiffWithXvec(c, dtype=dt).load_state_dict(flow_v1_state_dict(c))
    sr = int(g["sample_rate"])
    n_p, t1 = g["prompt_token"].shape[1], g["prompt_feat"].shape[1]
    kw = dict(prompt_token=g["prompt_token"], prompt_token_len=torch.tensor([n_p]), prompt_feat=g["prompt_feat"],
              prompt_feat_len=torch.tensor([t1]), embedding=g["embedding"], sample_rate=sr)
    cache = torch.zeros(1, 80, 0, 2)
    for i in (1, 2):
        tok = g[f"token{i}"]
        mel, cache = m.inference(token=tok, token_len=torch.tensor([tok.shape[1]]), flow_cache=cache, z=g[f"z{i}"], **kw)
        ref, ref_cache = g[f"mel{i}"], g[f"cache{i}"]
        assert mel.shape == ref.shape and cache.shape == ref_cache.shape
        assert torch.equal(cache[..., 0].cpu(), ref_cache[..., 0])                    # the noise half is copied, never computed
        cerr = (cache[..., 1].cpu() - ref_cache[..., 1]).abs().max().item()
        err = (mel.cpu() - ref).abs()
        print(f"v1 flow[{dt}] chunk {i}: mel L1 {err.mean():.3e} Linf {err.max():.3e}; mu-cache Linf {cerr:.3e}")
        assert err.mean().item() < l1 and err.max().item() < linf and cerr < linf
        cache = ref_cache.clone()     # chunk 2 starts from the reference's cache, so both chunks are compared like for like


def test_seeded_noise_and_graph(golden_dir):
    """Without an injected z the noise is torch.randn on the host generator (the reference's CPU draw under the same seed);
    the captured Euler loop reproduces the eager one bit for bit."""
    from cosyvoice_amd.flow_v1 import MaskedDiffWithXvec
    g = _golden(golden_dir)
    c = FlowV1Config.tiny()
    m = MaskedDiffWithXvec(c, dtype=torch.float16).load_state_dict(flow_v1_state_dict(c))
    kw = dict(token=g["token1"], token_len=torch.tensor([50]), prompt_token=g["prompt_token"], prompt_token_len=torch.tensor([12]),
              prompt_feat=g["prompt_feat"], prompt_feat_len=torch.tensor([20]), embedding=g["embedding"],
              flow_cache=torch.zeros(1, 80, 0, 2), sample_rate=int(g["sample_rate"]))
    torch.manual_seed(101)
    mel_a, cache_a = m.inference(**kw)
    assert torch.equal(cache_a[..., 0].cpu(), g["cache1"][..., 0])
    mel_b, _ = m.inference(z=g["z1"], **kw)
    assert torch.equal(mel_a, mel_b)
    m.decoder.use_graph = True
    m.inference(z=g["z1"], **kw)                       # capture
    mel_c, _ = m.inference(z=g["z1"], **kw)            # replay
    assert torch.equal(mel_c, mel_b)


def test_full_depth_vs_oracle():
    """FULL CosyVoice-v1 flow (6 conformer layers, 16 estimator stages x 4 transformer blocks, 10 Euler steps) against the
    oracle on the same injected noise: the north-star mel tolerance (L1 < 1e-3 in fp16) at an odd mel length."""
    from cosyvoice_amd.flow_v1 import MaskedDiffWithXvec
    from oracle import flow_v1 as o
    c = FlowV1Config.full()
    sd = flow_v1_state_dict(c)
    m = MaskedDiffWithXvec(c, dtype=torch.float16).load_state_dict(sd)
    g = torch.Generator().manual_seed(5)
    n_p, n_g, t1, sr = 30, 61, 52, 22050
    ptok = torch.randint(0, c.vocab_size, (1, n_p), generator=g, dtype=torch.int32)
    tok = torch.randint(0, c.vocab_size, (1, n_g), generator=g, dtype=torch.int32)
    pfeat = torch.clamp(torch.randn(1, t1, 80, generator=g) * 2 - 6, -11.5, 2.0)
    emb = torch.randn(1, c.spk_embed_dim, generator=g)
    t2 = m.mel_len(n_g, sr)
    z = torch.randn(1, 80, t1 + t2, generator=g)
    cache0 = torch.zeros(1, 80, 0, 2)
    with torch.no_grad():
        ref, ref_cache = o.inference(sd, c, tok, ptok, pfeat, emb, cache0, sr, z)
    mel, cache = m.inference(token=tok, token_len=torch.tensor([n_g]), prompt_token=ptok, prompt_token_len=torch.tensor([n_p]),
                             prompt_feat=pfeat, prompt_feat_len=torch.tensor([t1]), embedding=emb, flow_cache=cache0, sample_rate=sr, z=z)
    err = (mel.cpu() - ref).abs()
    print(f"v1 flow FULL depth fp16, T = {t1 + t2}: mel L1 {err.mean():.3e} Linf {err.max():.3e}")
    assert mel.shape == (1, 80, t2) and (t1 + t2) % 2 == 1
    assert err.mean().item() < 2.5e-3 and err.max().item() < 5e-2
    assert (cache.cpu() - ref_cache).abs().max().item() < 2e-2


def test_v1_stack_orchestrator(golden_dir):
    """The whole CosyVoice-v1 stack under CosyVoiceModel: vc() over 330 source tokens through MaskedDiffWithXvec (flow cache
    carried between chunks) + HiFT v1, streaming and not, against the chunk lengths the reference's own CosyVoiceModel yields
    around the reference's own v1 modules (golden); then tts() with the v1 TransformerLM in front."""
    from cosyvoice_amd.config import HiftConfig, TransformerLMConfig
    from cosyvoice_amd.flow_v1 import MaskedDiffWithXvec
    from cosyvoice_amd.hift import HiFTGenerator
    from cosyvoice_amd.llm_v1 import TransformerLM
    from cosyvoice_amd.model import CosyVoiceModel
    from cosyvoice_amd.weights import hift_state_dict, transformer_lm_state_dict
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, "v1_orchestrator_v1flow.npz")).items()}
    fc, hc, lc = FlowV1Config.tiny(), HiftConfig.v1(), TransformerLMConfig.tiny()
    flow = MaskedDiffWithXvec(fc, dtype=torch.float16).load_state_dict(flow_v1_state_dict(fc))
    hift = HiFTGenerator(hc, dtype=torch.float32).load_state_dict(hift_state_dict(hc))
    llm = TransformerLM(lc, dtype=torch.float16, max_len=1024).load_state_dict(transformer_lm_state_dict(lc))
    m = CosyVoiceModel(llm, flow, hift, fp16=False, sr=22050)
    assert m.mel_overlap_len == int(g["mel_overlap_len"]) and m.token_min_hop_len == int(g["token_min_hop_len"])
    args = (g["source_speech_token"], g["prompt_token"], g["prompt_feat"], g["embedding"])
    chunks = [o["tts_speech"] for o in m.vc(*args, stream=True)]
    assert [c.shape[1] for c in chunks] == g["stream_chunk_samples"].tolist()
    assert all(torch.isfinite(c).all() and c.abs().max() <= 0.99 + 1e-6 for c in chunks)
    full = [o["tts_speech"] for o in m.vc(*args, stream=False)]
    assert [c.shape[1] for c in full] == g["full_samples"].tolist()
    assert not m.flow_cache_dict and not m.mel_overlap_dict and not m.hift_cache_dict
    # text -> speech: the v1 LM feeds the same schedule
    gen = torch.Generator().manual_seed(1)
    text = torch.randint(0, lc.text_token_size, (1, 6), generator=gen)
    outs = [o["tts_speech"] for o in m.tts(text=text, flow_embedding=g["embedding"], llm_embedding=torch.randn(1, lc.spk_embed_dim, generator=gen),
                                           flow_prompt_speech_token=g["prompt_token"], prompt_speech_feat=g["prompt_feat"], stream=False)]
    n = outs[0].shape[1]
    assert len(outs) == 1 and n % 256 == 0 and torch.isfinite(outs[0]).all()
    assert flow.mel_len(2 * 6, 22050) <= n // 256 <= flow.mel_len(20 * 6, 22050)


def test_empty_prompt_and_short_chunk_vs_oracle():
    """Edge cases of flow.py:130-148 / length_regulator.py:49-70: no prompt at all (x1 empty, cond all zeros, cache = last 34
    frames only) and a chunk of <= 40 tokens (single interpolation), then a second call inheriting that cache."""
    from cosyvoice_amd.flow_v1 import MaskedDiffWithXvec
    from oracle import flow_v1 as o
    c = FlowV1Config.tiny()
    sd = flow_v1_state_dict(c)
    m = MaskedDiffWithXvec(c, dtype=torch.float16).load_state_dict(sd)
    g = torch.Generator().manual_seed(9)
    sr = 22050
    emb = torch.randn(1, c.spk_embed_dim, generator=g)
    ptok, pfeat = torch.zeros(1, 0, dtype=torch.int32), torch.zeros(1, 0, 80)
    cache = ref_cache = torch.zeros(1, 80, 0, 2)
    for n_g in (33, 47):
        tok = torch.randint(0, c.vocab_size, (1, n_g), generator=g, dtype=torch.int32)
        z = torch.randn(1, 80, m.mel_len(n_g, sr), generator=g)
        with torch.no_grad():
            ref, ref_cache = o.inference(sd, c, tok, ptok, pfeat, emb, ref_cache, sr, z)
        mel, cache = m.inference(token=tok, token_len=torch.tensor([n_g]), prompt_token=ptok, prompt_token_len=torch.tensor([0]),
                                 prompt_feat=pfeat, prompt_feat_len=torch.tensor([0]), embedding=emb, flow_cache=cache, sample_rate=sr, z=z)
        assert mel.shape == ref.shape and cache.shape == ref_cache.shape == (1, 80, 34, 2)
        err = (mel.cpu() - ref).abs()
        assert err.mean().item() < 2e-3 and err.max().item() < 2e-2
        assert torch.equal(cache[..., 0].cpu(), ref_cache[..., 0]) and (cache[..., 1].cpu() - ref_cache[..., 1]).abs().max().item() < 2e-2
        cache = ref_cache.clone()
