"""GPU: parity on the shapes the C4 bench line actually runs (VERDICT r02 "weak #1"): the bench teacher-forces its tokens, so a
numeric bug in these code paths cannot show in the bench itself.

  * LLM: fp16 Qwen2LM(max_batch=32, ctx_max=704) at 8 / 16 / 20 / 32 rows against the reference-minted long-context golden
    (tests/golden/llm_full_long.npz: the reference's own Qwen2LM.inference loop, llm/llm.py:823-874, prefill 282 + 250
    teacher-forced steps, context 282 -> 532) — covers the two-row-group skinny kernels at K = 896 / 4864, cv_rmsnorm_reduce as its
    own launch, cv_kv_retile at L = 282 and the second key tile per wave of the decode attention (context > 512).
  * cv_decode_attention alone at context {63, 64, 65, 127, 128, 511, 512, 513, 700} of ctx_max 704, fused and three-kernel forms,
    against fp32 torch.
  * flow: full depth at T = 1000 (250 + 250 tokens) against the reference mel (tests/golden/flow_long.npz, flow/flow.py:258-319).
  * HiFT decode at 500 (v2) / 861 (v1) frames against the reference waveform (tests/golden/hift_*_long.npz, hifigan/generator.py:349-381).

Tolerances are the stated ones: log-prob L-inf <= 5e-2, mel L1 <= 1e-3, fp32 waveform L-inf <= 1e-4."""
import os

import numpy as np
import pytest
import torch

from cosyvoice_amd.config import FlowConfig, HiftConfig, LlmConfig
from cosyvoice_amd.weights import flow_state_dict, hift_state_dict, llm_state_dict

pytestmark = pytest.mark.gpu


def _load(golden_dir, name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, name + ".npz")).items()}


@pytest.fixture(scope="module")
def long_lm():
    from cosyvoice_amd.llm import Qwen2LM
    cfg = LlmConfig.full()
    return Qwen2LM(cfg, dtype=torch.float16, max_batch=32, ctx_max=704, max_out=512).load_state_dict(llm_state_dict(cfg))


@pytest.mark.parametrize("rows", [8, 16, 20, 32])
def test_llm_long_context_rows_vs_reference_loop(golden_dir, long_lm, rows):
    """Every batch row at every stored step within the stated 5e-2 of the reference's fp32 log-probs."""
    g = _load(golden_dir, "llm_full_long")
    keep = g["rows"].tolist()
    lp = long_lm.forced_logits(g["text"], g["prompt_text"], g["prompt_speech"], g["forced"].tolist(), rows=rows, keep=keep).cpu()
    ref = g["logps"]
    assert lp.shape == (len(keep), rows, ref.shape[1])
    d = (lp - ref[:, None]).abs()
    per_step = d.amax(dim=(1, 2))
    print(f"llm long context, {rows} rows fp16: logp Linf per stored step {dict(zip(keep, [round(v, 4) for v in per_step.tolist()]))}; "
          f"rows differ among themselves by {(lp - lp[:, :1]).abs().max().item():.2e}; "
          f"argmax agreement {(lp.argmax(-1) == ref.argmax(-1)[:, None]).float().mean().item():.3f}")
    assert torch.isfinite(lp).all()
    assert per_step.max().item() < 5e-2
    # a row's result must not depend on which MFMA row group / batch slot it sits in beyond fp16 rounding of different tile shapes
    assert (lp - lp[:, :1]).abs().max().item() < 2e-2


@pytest.mark.parametrize("dt,tol", [(torch.float16, 4e-3), (torch.bfloat16, 2e-2)])
def test_decode_attention_long_contexts(dt, tol):
    """cv_decode_attention at contexts around every 64-key tile edge up to ctx_max 704 (1 .. 11 tiles: every wave's first tile, the
    wid + 8 second tile, the log-sum-exp merge over all 8 waves), fused (RoPE + append on fragment-tiled caches) and three-kernel
    (row-major caches) forms, against fp32 torch on the same 16-bit cache contents."""
    from cosyvoice_amd import ops
    torch.manual_seed(11)
    dev = "cuda"
    Hq, Hkv, ctx_max = 14, 2, 704
    ctxs = [63, 64, 65, 127, 128, 511, 512, 513, 700]
    B = len(ctxs)
    qkv_dim = (Hq + 2 * Hkv) * 64
    inv = (1.0 / (1e6 ** (torch.arange(0, 64, 2, dtype=torch.float32) / 64))).to(dev)
    kc = torch.zeros(B, Hkv, ctx_max, 64, device=dev, dtype=dt)
    vc = torch.zeros(B, Hkv, 64, ctx_max, device=dev, dtype=dt)
    ctx0 = torch.tensor(ctxs + [0] * (16 - B), device=dev, dtype=torch.int32)
    for b, n in enumerate(ctxs):
        kc[b, :, :n] = torch.randn(Hkv, n, 64, device=dev).to(dt)
        vc[b, :, :, :n] = torch.randn(Hkv, 64, n, device=dev).to(dt)
        # keys beyond the context must never be read as data: poison them
        kc[b, :, n + 1:] = 1e4
        vc[b, :, :, n + 1:] = 1e4
    kc_hist, vc_hist = kc.clone(), vc.clone()
    qkv = torch.randn(16, qkv_dim, device=dev)
    q = torch.zeros(16, Hq * 64, device=dev, dtype=dt)
    ops.rope_append(qkv, ctx0, B, 1, Hq, Hkv, inv, q, kc, vc, ctx_max)
    out = torch.zeros(16, Hq * 64, device=dev, dtype=dt)
    ops.decode_attention(q, kc, vc, ctx0, 1, out, B, Hq, Hkv, ctx_max, 0.125)
    # fused form on fragment-tiled caches
    kt, vt = torch.zeros_like(kc), torch.zeros_like(vc)
    ops.kv_retile(kc_hist, vc_hist, kt, vt, B, Hkv, ctx_max, ctx_max)
    ang = torch.arange(ctx_max, dtype=torch.float32, device=dev)[:, None] * inv[None, :]
    tab = torch.cat([ang.cos(), ang.sin()], 1).contiguous()
    out2 = torch.zeros_like(out)
    ops.decode_attention(q, kt, vt, ctx0, 1, out2, B, Hq, Hkv, ctx_max, 0.125, qkv=qkv, inv_freq=tab)
    torch.cuda.synchronize()
    worst = [0.0, 0.0]
    for b, n in enumerate(ctxs):
        K = kc[b, :, :n + 1].float().repeat_interleave(Hq // Hkv, 0)
        V = vc[b, :, :, :n + 1].float().transpose(1, 2).repeat_interleave(Hq // Hkv, 0)
        s = torch.einsum("hd,hnd->hn", q[b].float().view(Hq, 64), K) * 0.125
        o = torch.einsum("hn,hnd->hd", torch.softmax(s, -1), V)
        e1 = (out[b].float().view(Hq, 64) - o).abs().max().item()
        e2 = (out2[b].float().view(Hq, 64) - o).abs().max().item()
        worst = [max(worst[0], e1), max(worst[1], e2)]
        assert e1 < tol and e2 < tol, (n, e1, e2)
    print(f"decode attention [{dt}] ctx {ctxs}: Linf three-kernel {worst[0]:.2e}, fused {worst[1]:.2e}")
    # the fused form appended the same K row / V column to the tiled caches
    kt_ref, vt_ref = torch.zeros_like(kc), torch.zeros_like(vc)
    ops.kv_retile(kc, vc, kt_ref, vt_ref, B, Hkv, ctx_max, ctx_max)
    torch.cuda.synchronize()
    for b, n in enumerate(ctxs):
        idx_k = [ops.kv_tile_index(ctx_max, n, d) for d in range(64)]
        idx_v = [ops.kv_tile_index(ctx_max, n, d, v=True) for d in range(64)]
        assert (kt.view(B, Hkv, -1)[b][:, idx_k].float() - kt_ref.view(B, Hkv, -1)[b][:, idx_k].float()).abs().max().item() < 2e-2
        assert torch.equal(vt.view(B, Hkv, -1)[b][:, idx_v], vt_ref.view(B, Hkv, -1)[b][:, idx_v])


def test_full_depth_flow_T1000_vs_reference_mel(golden_dir):
    """The flow of one C4 utterance (250 prompt + 250 generated tokens -> T = 1000, chunk mask 50, 56 blocks x 10 CFG Euler steps),
    fp16 operands, against the reference's fp32 mel: north_star mel L1 <= 1e-3."""
    from cosyvoice_amd.flow import CausalMaskedDiffWithXvec
    cfg = FlowConfig.full()
    g = _load(golden_dir, "flow_long")
    flow = CausalMaskedDiffWithXvec(cfg, dtype=torch.float16).load_state_dict(flow_state_dict(cfg))
    flow.encoder.static_chunk_size = 50
    mel = flow.inference_batch(g["token"], g["prompt_token"], g["prompt_feat"], g["embedding"]).cpu()
    assert mel.shape == (1, 80, 500)
    d = (mel[:, :, ::4] - g["mel_sub4"]).abs()
    ca = (mel[0].abs().mean(dim=1) - g["mel_chan_absmean"]).abs().max().item()
    fm = (mel[0].mean(dim=0) - g["mel_frame_mean"]).abs()
    print(f"flow T=1000 vs reference [fp16]: mel L1 {d.mean().item():.3e} Linf {d.max().item():.3e} (every 4th frame); per-channel abs-mean "
          f"diff {ca:.2e}; per-frame mean diff max {fm.max().item():.2e} (all 500 frames)")
    assert d.mean().item() < 1e-3 and d.max().item() < 1.5e-2          # the stated tolerance (mel L1) + the L-inf bound of the T <= 500 goldens
    # diagnostics over ALL 500 frames (the exact mel is stored for every 4th only): no channel / frame drifts by more than a few L1's
    assert ca < 3e-3 and fm.max().item() < 2e-3
    # the batched launch shapes of the bench (8 utterances = 16 CFG rows): utterance 0 of a batch of 8 equals the single run
    rep = lambda t: t.repeat(8, *([1] * (t.dim() - 1)))
    mel8 = flow.inference_batch(rep(g["token"]), rep(g["prompt_token"]), rep(g["prompt_feat"]), rep(g["embedding"])).cpu()
    d8 = (mel8[:, :, ::4] - g["mel_sub4"]).abs()
    print(f"flow T=1000, batch 8: mel L1 {d8.mean().item():.3e} Linf {d8.max().item():.3e}; rows differ by {(mel8 - mel8[:1]).abs().max().item():.2e}")
    assert d8.mean(dim=(1, 2)).max().item() < 1e-3 and d8.max().item() < 1.5e-2


def _synth_mel(batch, frames, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.clamp(torch.randn(batch, 80, frames, generator=g) * 2.0 - 6.0, -11.5, 2.0)


@pytest.mark.parametrize("tag,cfg", [("v2", HiftConfig.v2()), ("v1", HiftConfig.v1())])
@pytest.mark.parametrize("mode,tol", [("bf16x3", 1e-4), ("exact", 1e-4)])
def test_hift_decode_baseline_length_vs_reference(golden_dir, tag, cfg, mode, tol):
    """HiFTGenerator.decode at 500 frames (v2, 240 000 samples) / 861 frames (v1, 220 416 samples), fp32 tensors, bf16x3 (the bench's
    default) and exact-f32 products, against the reference waveform: L-inf <= 1e-4 at +-0.99 full scale.  Batch of 8 == single."""
    from cosyvoice_amd.hift import HiFTGenerator
    g = _load(golden_dir, f"hift_{tag}_long")
    frames = int(g["frames"])
    mel = _synth_mel(1, frames, int(g["mel_seed"]))
    s = torch.randn(1, 1, frames * cfg.total_upsample, generator=torch.Generator().manual_seed(int(g["s_seed"]))) * 0.05
    m = HiFTGenerator(cfg, dtype=torch.float32, f32_products=mode).load_state_dict(hift_state_dict(cfg))
    wav = m.decode(mel.cuda(), s.cuda()).cpu()
    assert wav.shape[1] == int(g["n_samples"])
    e = (wav[:, ::8] - g["wav_sub8"]).abs().max().item()
    blk = wav[0, : wav.shape[1] // 2400 * 2400].view(-1, 2400).abs()
    eb = (blk.max(dim=1).values - g["wav_block_absmax"]).abs().max().item()
    em = (blk.mean(dim=1) - g["wav_block_absmean"]).abs().max().item()
    print(f"hift {tag} {frames} frames [{mode}]: wav Linf {e:.2e} (every 8th sample), block abs-max diff {eb:.2e}, block abs-mean diff {em:.2e}")
    assert e < tol and eb < tol and em < tol
    f0 = m.f0_predictor(mel.cuda()).cpu()
    assert (f0 - g["f0"]).abs().max().item() < 2e-3
    wav8 = m.decode(mel.cuda().repeat(8, 1, 1), s.cuda().repeat(8, 1, 1)).cpu()
    assert (wav8 - wav).abs().max().item() < 1e-6
