"""GPU: LLM decode kernels and the Qwen2LM host path (through the C ABI) against torch references, the
reference-minted golden (teacher-forced log-probs) and the CPU oracle (token-exact with injected uniforms)."""
import math
import os

import numpy as np
import pytest
import torch

from cosyvoice_amd.config import LlmConfig
from cosyvoice_amd.weights import llm_state_dict

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,ksplit", [(1, 896, 896, 1), (8, 1152, 896, 1), (8, 896, 4864, 4), (3, 6564, 896, 1), (16, 256, 512, 2),
                                          (8, 896, 4864, 2), (4, 320, 4864, 1)])  # last: K slice beyond 16 k-steps per wave
def test_skinny_gemm_modes(dt, M, N, K, ksplit):
    from cosyvoice_amd import ops
    torch.manual_seed(0)
    dev = "cuda"
    A = torch.zeros(16, K, device=dev, dtype=dt)
    A[:M] = torch.randn(M, K, device=dev).to(dt)
    W = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt)
    bias = torch.randn(N, device=dev)
    Wp = ops.pack_skinny(W)
    ref = A[:M].float() @ W.float().t()
    Np = (N + 15) // 16 * 16
    slabs = torch.zeros(ksplit, 16, Np, device=dev)
    ops.skinny_gemm(A, Wp, M, N, K, bias=bias, ksplit=ksplit, out_f32=slabs, ldo=Np, slab_stride=16 * Np)
    torch.cuda.synchronize()
    got = slabs.sum(0)[:M, :N]
    tol = 2e-2 if dt == torch.bfloat16 else 3e-3
    assert (got - (ref + bias)).abs().max().item() < tol * max(1.0, ref.abs().max().item())
    if M < 16:
        assert slabs[:, M:].abs().max().item() == 0.0  # rows >= M untouched
    if ksplit == 1:
        x = torch.randn(16, Np, device=dev)
        x0 = x.clone()
        ops.skinny_gemm(A, Wp, M, N, K, mode=1, out_f32=x, ldo=Np)
        torch.cuda.synchronize()
        assert (x[:M, :N] - (x0[:M, :N] + ref)).abs().max().item() < tol * max(1.0, ref.abs().max().item())
    # capped grids (workgroups walk several tile groups, next group's weights prefetched) are bit-identical
    for cap in (7 * ksplit, 2 * ksplit, 40):
        s2 = torch.zeros_like(slabs)
        ops.skinny_gemm(A, Wp, M, N, K, bias=bias, ksplit=ksplit, out_f32=s2, ldo=Np, slab_stride=16 * Np, max_wgs=cap)
        torch.cuda.synchronize()
        assert torch.equal(s2, slabs), cap
        if ksplit == 1:
            x2 = x0.clone()
            ops.skinny_gemm(A, Wp, M, N, K, mode=1, out_f32=x2, ldo=Np, max_wgs=cap)
            torch.cuda.synchronize()
            assert torch.equal(x2, x), cap


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_skinny_swiglu(dt):
    from cosyvoice_amd import ops
    torch.manual_seed(1)
    dev = "cuda"
    M, K, I = 5, 896, 4864
    A = torch.zeros(16, K, device=dev, dtype=dt)
    A[:M] = torch.randn(M, K, device=dev).to(dt)
    g = (torch.randn(I, K, device=dev) / K ** 0.5).to(dt)
    u = (torch.randn(I, K, device=dev) / K ** 0.5).to(dt)
    Wp = ops.pack_skinny(torch.cat([g, u], 0).contiguous(), interleave=True)
    h = torch.zeros(16, I, device=dev, dtype=dt)
    ops.skinny_gemm(A, Wp, M, 2 * I, K, mode=2, out_act=h, ldoa=I)
    torch.cuda.synchronize()
    ref = torch.nn.functional.silu(A[:M].float() @ g.float().t()) * (A[:M].float() @ u.float().t())
    tol = 3e-2 if dt == torch.bfloat16 else 4e-3
    assert (h[:M].float() - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item())
    for cap in (3, 96, 192):
        h2 = torch.zeros_like(h)
        ops.skinny_gemm(A, Wp, M, 2 * I, K, mode=2, out_act=h2, ldoa=I, max_wgs=cap)
        torch.cuda.synchronize()
        assert torch.equal(h2, h), cap


def test_rmsnorm_reduce_and_rope_decode_attention():
    from cosyvoice_amd import ops
    torch.manual_seed(2)
    dev, dt = "cuda", torch.bfloat16
    B, H, Hq, Hkv, ctx_max = 3, 896, 14, 2, 128
    x = torch.randn(16, H, device=dev)
    slabs = torch.randn(4, 16, H, device=dev)
    gam = torch.randn(H, device=dev)
    xn = torch.zeros(16, H, device=dev, dtype=dt)
    x0 = x.clone()
    ops.rmsnorm_reduce(x, gam, 1e-6, xn, B, slabs=slabs, nslab=4, slab_stride=16 * H, ld_slab=H)
    torch.cuda.synchronize()
    xs = x0[:B] + slabs[:, :B].sum(0)
    assert (x[:B] - xs).abs().max().item() < 1e-5
    ref = gam * xs * torch.rsqrt(xs.pow(2).mean(-1, keepdim=True) + 1e-6)
    assert (xn[:B].float() - ref).abs().max().item() < 3e-2
    # rope + append + decode attention against an fp32 reference built from the same 16-bit cache contents
    qkv_dim = (Hq + 2 * Hkv) * 64
    inv = (1.0 / (1e6 ** (torch.arange(0, 64, 2, dtype=torch.float32) / 64))).to(dev)
    kc = torch.zeros(B, Hkv, ctx_max, 64, device=dev, dtype=dt)
    vc = torch.zeros(B, Hkv, 64, ctx_max, device=dev, dtype=dt)
    ctx0 = torch.tensor([70, 5, 100] + [0] * 13, device=dev, dtype=torch.int32)
    # history
    for b in range(B):
        n = int(ctx0[b])
        kc[b, :, :n] = torch.randn(Hkv, n, 64, device=dev).to(dt)
        vc[b, :, :, :n] = torch.randn(Hkv, 64, n, device=dev).to(dt)
    qkv = torch.randn(16, qkv_dim, device=dev)
    q = torch.zeros(16, Hq * 64, device=dev, dtype=dt)
    ops.rope_append(qkv, ctx0, B, 1, Hq, Hkv, inv, q, kc, vc, ctx_max)
    out = torch.zeros(16, Hq * 64, device=dev, dtype=dt)
    ops.decode_attention(q, kc, vc, ctx0, 1, out, B, Hq, Hkv, ctx_max, 0.125)
    torch.cuda.synchronize()

    def rope(v, pos):
        ang = pos * inv
        c, s = torch.cat([ang.cos(), ang.cos()]), torch.cat([ang.sin(), ang.sin()])
        rot = torch.cat([-v[..., 32:], v[..., :32]], -1)
        return v * c + rot * s

    for b in range(B):
        n = int(ctx0[b])
        qr = rope(qkv[b, :Hq * 64].view(Hq, 64), float(n))
        kr = rope(qkv[b, Hq * 64:(Hq + Hkv) * 64].view(Hkv, 64), float(n))
        assert (q[b].float().view(Hq, 64) - qr).abs().max().item() < 3e-2
        assert (kc[b, :, n].float() - kr).abs().max().item() < 3e-2
        assert (vc[b, :, :, n].float() - qkv[b, (Hq + Hkv) * 64:].view(Hkv, 64)).abs().max().item() < 3e-2
        K = kc[b, :, :n + 1].float().repeat_interleave(Hq // Hkv, 0)
        V = vc[b, :, :, :n + 1].float().transpose(1, 2).repeat_interleave(Hq // Hkv, 0)
        s = torch.einsum("hd,hnd->hn", q[b].float().view(Hq, 64), K) * 0.125
        o = torch.einsum("hn,hnd->hd", torch.softmax(s, -1), V)
        assert (out[b].float().view(Hq, 64) - o).abs().max().item() < 2e-2
    # fused form: RoPE + append + attention in one launch must reproduce the three-kernel result
    kc2, vc2 = kc.clone(), vc.clone()
    for b in range(B):
        kc2[b, :, int(ctx0[b])] = 0
        vc2[b, :, :, int(ctx0[b])] = 0
    out2 = torch.zeros_like(out)
    ang = torch.arange(ctx_max, dtype=torch.float32, device=dev)[:, None] * inv[None, :]
    tab = torch.cat([ang.cos(), ang.sin()], 1).contiguous()
    # the fused form reads and appends to FRAGMENT-TILED caches (cv_kv_retile)
    kt2, vt2 = torch.zeros_like(kc2), torch.zeros_like(vc2)
    ops.kv_retile(kc2, vc2, kt2, vt2, B, Hkv, ctx_max, ctx_max)
    ops.decode_attention(q, kt2, vt2, ctx0, 1, out2, B, Hq, Hkv, ctx_max, 0.125, qkv=qkv, inv_freq=tab)
    torch.cuda.synchronize()
    assert (out2[:B].float() - out[:B].float()).abs().max().item() < 1e-2
    # tiled caches after the append == re-tiled row-major caches that hold the three-kernel path's new K / V
    kt_ref, vt_ref = torch.zeros_like(kc), torch.zeros_like(vc)
    ops.kv_retile(kc, vc, kt_ref, vt_ref, B, Hkv, ctx_max, ctx_max)
    torch.cuda.synchronize()
    assert (kt2[:B].float() - kt_ref[:B].float()).abs().max().item() < 2e-2 and torch.equal(vt2[:B], vt_ref[:B])
    # and the documented index map: element (key, d) of the row-major cache sits at kv_tile_index(key, d)
    flat_k, flat_v = kt_ref.view(B, Hkv, -1), vt_ref.view(B, Hkv, -1)
    for key, d in ((0, 0), (5, 63), (70, 17), (100, 40), (127, 31)):
        assert torch.equal(flat_k[:, :, ops.kv_tile_index(ctx_max, key, d)], kc[:, :, key, d])
        assert torch.equal(flat_v[:, :, ops.kv_tile_index(ctx_max, key, d, v=True)], vc[:, :, d, key])
    # fused RMSNorm prologue of the skinny GEMM == rmsnorm_reduce + plain skinny GEMM
    torch.manual_seed(5)
    W = (torch.randn(1152, H, device=dev) / H ** 0.5).to(dt)
    Wp = ops.pack_skinny(W)
    xin = torch.randn(16, H, device=dev)
    xin[B:] = 0
    xo = torch.zeros(16, H, device=dev)
    o_f = torch.zeros(16, 1152, device=dev)
    ops.skinny_gemm(xn, Wp, B, 1152, H, out_f32=o_f, ldo=1152,
                    norm=dict(x=xin, gamma=gam, eps=1e-6, x_out=xo, slabs=slabs, nslab=4, slab_stride=16 * H, ld_slab=H))
    xr = xin.clone()
    xn2 = torch.zeros(16, H, device=dev, dtype=dt)
    ops.rmsnorm_reduce(xr, gam, 1e-6, xn2, B, slabs=slabs, nslab=4, slab_stride=16 * H, ld_slab=H)
    o_r = torch.zeros(16, 1152, device=dev)
    ops.skinny_gemm(xn2, Wp, B, 1152, H, out_f32=o_r, ldo=1152)
    torch.cuda.synchronize()
    assert (xo[:B] - xr[:B]).abs().max().item() < 1e-5
    assert (o_f[:B] - o_r[:B]).abs().max().item() < 2e-2


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M", [3, 8, 16])
def test_split_rmsnorm_producer_consumer(dt, M):
    """RMSNorm split over two skinny launches (the decode step's o_proj -> gate/up pair): the in-place-residual producer also
    emits the updated rows as 16-bit and per-workgroup partial sums of squares; the SwiGLU consumer reads those rows without a
    prologue and applies 1/rms in its epilogue, gamma folded into the packed weights.  Against torch, and against the
    fused-prologue form of the same kernel."""
    from cosyvoice_amd import ops
    torch.manual_seed(M)
    dev = "cuda"
    H, Q, I, eps = 896, 896, 4864, 1e-6
    ao = torch.zeros(16, Q, device=dev, dtype=dt)
    ao[:M] = torch.randn(M, Q, device=dev).to(dt)
    wo = (torch.randn(H, Q, device=dev) / Q ** 0.5).to(dt)
    x0 = torch.randn(16, H, device=dev) * 3
    x = x0.clone()
    xb = torch.zeros(16, H, device=dev, dtype=dt)
    ssp = torch.full((H // 16, 16), float("nan"), device=dev)
    ops.skinny_gemm(ao, ops.pack_skinny(wo), M, H, Q, mode=1, out_f32=x, ldo=H, split_out=dict(xb=xb, ss=ssp))
    x_ref = x0[:M] + ao[:M].float() @ wo.float().t()
    assert (x[:M] - x_ref).abs().max().item() < 1e-3 * x_ref.abs().max().item()
    assert torch.equal(xb[:M], x[:M].to(dt)) and torch.equal(x[M:], x0[M:])          # 16-bit copy of exactly what was stored
    ss = ssp[:, :M].sum(0)
    assert torch.isfinite(ss).all() and ((ss - (x[:M] ** 2).sum(1)).abs() / ss).max().item() < 1e-5
    gamma = 1 + 0.1 * torch.randn(H, device=dev)
    g = (torch.randn(I, H, device=dev) / H ** 0.5).to(dt)
    u = (torch.randn(I, H, device=dev) / H ** 0.5).to(dt)
    wp_g = ops.pack_skinny(torch.cat([g.float() * gamma, u.float() * gamma], 0).to(dt).contiguous(), interleave=True)
    h = torch.zeros(16, I, device=dev, dtype=dt)
    ops.skinny_gemm(xb, wp_g, M, 2 * I, H, mode=2, out_act=h, ldoa=I, split_in=dict(rs=ssp, n=H // 16, eps=eps))
    xn = x[:M] * torch.rsqrt((x[:M] ** 2).mean(1, keepdim=True) + eps) * gamma
    ref = torch.nn.functional.silu(xn @ g.float().t()) * (xn @ u.float().t())
    tol = 4e-2 if dt == torch.bfloat16 else 5e-3
    scale = max(1.0, ref.abs().max().item())
    assert (h[:M].float() - ref).abs().max().item() < tol * scale
    # the fused-prologue form of the same product
    h2 = torch.zeros_like(h)
    ops.skinny_gemm(xb, ops.pack_skinny(torch.cat([g, u], 0).contiguous(), interleave=True), M, 2 * I, H, mode=2, out_act=h2, ldoa=I,
                    norm=dict(x=x, gamma=gamma, eps=eps))
    assert (h2[:M].float() - h[:M].float()).abs().max().item() < tol * scale
    assert (h[M:] == 0).all()


def test_decode_step_split_norm_equals_fused_prologue(golden_dir):
    """Qwen2LM decode loop with the post-attention RMSNorm split over o_proj / gate-up (default) against the fused-prologue
    form: both within the golden tolerance, and close to each other."""
    from cosyvoice_amd.llm import Qwen2LM
    cfg = LlmConfig.tiny()
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, "llm_tiny.npz")).items()}
    lm = Qwen2LM(cfg, dtype=torch.float16, max_batch=4, ctx_max=256, max_out=256).load_state_dict(llm_state_dict(cfg))
    out = {}
    for split in (True, False):
        lm.split_norm = split
        out[split] = lm.forced_logits(g["text"], g["prompt_text"], g["prompt_speech"], g["forced"].tolist()).cpu()
        assert (out[split] - g["logps"]).abs().max().item() < 2e-2
    assert (out[True] - out[False]).abs().max().item() < 2e-2 and not torch.equal(out[True], out[False])


@pytest.mark.parametrize("dt,tol", [(torch.bfloat16, 1.5e-1), (torch.float16, 2e-2)])
def test_teacher_forced_logp_vs_golden(golden_dir, dt, tol):
    from cosyvoice_amd.llm import Qwen2LM
    cfg = LlmConfig.tiny()
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, "llm_tiny.npz")).items()}
    lm = Qwen2LM(cfg, dtype=dt, max_batch=4, ctx_max=256, max_out=256).load_state_dict(llm_state_dict(cfg))
    for use_graph in (False, True):
        lm.use_graph = use_graph
        lp = lm.forced_logits(g["text"], g["prompt_text"], g["prompt_speech"], g["forced"].tolist()).cpu()
        err = (lp - g["logps"]).abs().max().item()
        print(f"llm logp[{dt}, graph={use_graph}] Linf {err:.3e}")
        assert err < tol


def test_sampler_kernel_matches_oracle_on_identical_logits():
    """cv_sample_ras against the oracle's ras_sampling / sampling_ids on the SAME fp32 logits with injected uniforms:
    nucleus pick, repetition fallback (random_sampling over the full distribution), EOS redraw before min_len,
    ids above EOS skipped, forced override."""
    import ctypes as C
    from cosyvoice_amd import _lib as L
    from cosyvoice_amd import ops
    from oracle import llm as ol
    dev = "cuda"
    V, eos, B, H = 6564, 6561, 8, 64
    g = torch.Generator().manual_seed(11)
    logits = (torch.randn(B, 6576, generator=g) * 3.0)
    logits[3, eos] = 30.0        # sequence 3: EOS dominates -> must be redrawn while step < min_len
    logits[4, eos + 1] = 30.0    # sequence 4: a fill token (> EOS) dominates -> skipped, nothing emitted
    hist_len = 6
    hist = torch.randint(0, 6561, (B, hist_len), generator=g, dtype=torch.int32)
    # sequence 1 and 2: make the nucleus pick a token already in the window -> fallback path
    top1 = logits[:, :V].argmax(-1)
    hist[1, -1] = top1[1]
    hist[2, 2] = top1[2]
    logits[1, top1[1]] += 6.0
    logits[2, top1[2]] += 6.0
    uni = torch.rand(B, 101, 2, generator=g) * 0.98
    emb = torch.randn(V, H, generator=g)
    st = dict(step=torch.full((B,), 3, dtype=torch.int32), pos=torch.full((B,), 50, dtype=torch.int32),
              n_emitted=torch.full((B,), hist_len, dtype=torch.int32), finished=torch.zeros(B, dtype=torch.int32),
              min_len=torch.full((B,), 10, dtype=torch.int32), max_len=torch.full((B,), 100, dtype=torch.int32))
    st["min_len"][5] = 0  # sequence 5: past min_len (EOS allowed)
    out_tokens = torch.zeros(B, 32, dtype=torch.int32)
    out_tokens[:, :hist_len] = hist
    d = {k: v.to(dev) for k, v in st.items()}
    lg_d, uni_d, emb_d, out_d = logits.to(dev), uni.to(dev), emb.to(dev), out_tokens.to(dev)
    x_d = torch.zeros(B, H, device=dev)
    p = L.SampleParams()
    p.logits, p.ldl, p.V, p.B = lg_d.data_ptr(), 6576, V, B
    p.eos, p.top_k, p.top_p, p.win_size, p.tau_r = eos, 25, 0.8, 10, 0.1
    p.seed, p.uniforms, p.max_trials = 0, uni_d.data_ptr(), 100
    p.min_len, p.max_len = d["min_len"].data_ptr(), d["max_len"].data_ptr()
    p.forced, p.forced_ld = None, 0
    p.step, p.pos, p.n_emitted, p.finished = d["step"].data_ptr(), d["pos"].data_ptr(), d["n_emitted"].data_ptr(), d["finished"].data_ptr()
    p.out_tokens, p.out_ld = out_d.data_ptr(), 32
    p.emb_table, p.emb_dim, p.x, p.ldx = emb_d.data_ptr(), H, x_d.data_ptr(), H
    ops.sample_ras(p)
    torch.cuda.synchronize()
    ne, fin, toks, xs = d["n_emitted"].cpu(), d["finished"].cpu(), out_d.cpu(), x_d.cpu()
    n_fallback = 0
    for b in range(B):
        it = iter(uni[b].tolist())
        ignore_eos = 3 < int(st["min_len"][b])
        lp = logits[b, :V].log_softmax(-1)
        try:
            ref = ol.sampling_ids(lp, hist[b].tolist(), ignore_eos, eos, lambda t: tuple(uni[b, t].tolist()))
        except RuntimeError:
            assert int(fin[b]) == 3
            continue
        pc, ic = ol.nucleus_candidates(lp)
        first = int(ic[ol._inverse_cdf(pc, uni[b, 0, 0].item())])
        n_fallback += int(first in hist[b].tolist()[-10:])
        assert int(d["pos"][b].item()) == 51 and int(d["step"][b].item()) == 4
        if ref == eos:
            assert int(fin[b]) == 1 and int(ne[b]) == hist_len
        elif ref > eos:
            assert int(fin[b]) == 0 and int(ne[b]) == hist_len  # skipped (llm.py:869-870)
        else:
            assert int(ne[b]) == hist_len + 1 and int(toks[b, hist_len]) == ref, (b, ref, int(toks[b, hist_len]))
            assert torch.equal(xs[b], emb[ref])
    assert n_fallback >= 2  # the repetition-aware branch was exercised


def test_generation_first_tokens_match_oracle():
    """Model-level: with weights rounded to bf16 on both sides and moderate injected uniforms the first sampled token of
    every sequence (prefill logits -> sampler) equals the oracle's; later tokens may diverge at bf16 near-ties, so the
    agreement of the common prefix is only reported."""
    from cosyvoice_amd.llm import Qwen2LM
    from oracle import llm as ol
    cfg = LlmConfig.tiny()
    sd = llm_state_dict(cfg, round_to=torch.bfloat16)
    lm = Qwen2LM(cfg, dtype=torch.bfloat16, max_batch=4, ctx_max=256, max_out=256).load_state_dict(sd)
    g = torch.Generator().manual_seed(3)
    B = 3
    texts = [torch.randint(0, cfg.vocab_size, (1, 5), generator=g, dtype=torch.int32) for _ in range(B)]
    ptexts = [torch.randint(0, cfg.vocab_size, (1, 3), generator=g, dtype=torch.int32) for _ in range(B)]
    pspeech = [torch.randint(0, cfg.speech_token_size, (1, 7), generator=g, dtype=torch.int32) for _ in range(B)]
    # per-redraw uniforms (the same table is consumed at every step, indexed by the redraw number, on both sides);
    # nucleus uniforms kept below 0.6 so the pick is not in the near-tie tail of the candidate list
    uni = torch.zeros(16, 101, 2)
    uni[:, :, 0] = torch.rand(16, 101, generator=g) * 0.6
    uni[:, :, 1] = torch.rand(16, 101, generator=g)
    got = lm.generate_batch(texts, ptexts, pspeech, uniforms=uni)
    for b in range(B):
        ref = list(ol.lm_inference(sd, cfg, texts[b], ptexts[b], pspeech[b], uniforms=lambda t: tuple(uni[b, t].tolist())))
        n = min(len(ref), len(got[b]))
        first_diff = next((i for i in range(n) if ref[i] != got[b][i]), n)
        print(f"seq {b}: oracle {len(ref)} tokens, hip {len(got[b])} tokens, exact prefix {first_diff}")
        assert n > 0 and ref[0] == got[b][0]
        assert 10 <= len(got[b]) <= 100  # min/max token-text ratios respected


def test_inference_without_prompt_and_single_text_token():
    """Edge cases of Qwen2LM.inference (llm.py:823-874): empty prompt text AND empty prompt speech (zero-shot without a
    prompt: lm_input = [sos, text, task_id]) and a one-token text (min_len 2, max_len 20).  The generator yields python
    ints within the token-text ratio bounds; the first token equals the oracle's."""
    from cosyvoice_amd.llm import Qwen2LM
    from oracle import llm as ol
    cfg = LlmConfig.tiny()
    sd = llm_state_dict(cfg, round_to=torch.bfloat16)
    lm = Qwen2LM(cfg, dtype=torch.bfloat16, max_batch=4, ctx_max=256, max_out=256).load_state_dict(sd)
    g = torch.Generator().manual_seed(5)
    uni = torch.zeros(16, 101, 2)
    uni[:, :, 0] = torch.rand(16, 101, generator=g) * 0.6
    uni[:, :, 1] = torch.rand(16, 101, generator=g)
    empty_i = torch.zeros(1, 0, dtype=torch.int32)
    for n_text in (6, 1):
        text = torch.randint(0, cfg.vocab_size, (1, n_text), generator=g, dtype=torch.int32)
        got = lm.generate_batch([text], [empty_i], [empty_i], uniforms=uni)[0]
        ref = list(ol.lm_inference(sd, cfg, text, empty_i, empty_i, uniforms=lambda t: tuple(uni[0, t].tolist())))
        assert 2 * n_text <= len(got) <= 20 * n_text
        assert all(isinstance(t, int) and 0 <= t < cfg.speech_token_size for t in got)
        assert got[0] == ref[0]
        # the reference-signature generator gives the same stream of ints
        lm.seed = 0
        toks = list(lm.inference(text=text.cuda(), text_len=torch.tensor([n_text], dtype=torch.int32), prompt_text=empty_i.cuda(),
                                 prompt_text_len=torch.tensor([0], dtype=torch.int32), prompt_speech_token=empty_i.cuda(),
                                 prompt_speech_token_len=torch.tensor([0], dtype=torch.int32), embedding=torch.zeros(0, 192)))
        assert 2 * n_text <= len(toks) <= 20 * n_text and all(isinstance(t, int) for t in toks)


def test_ragged_batch_equals_single_sequences():
    """generate_batch over sequences of different text / prompt lengths (left-aligned slots, per-sequence attention length
    and decode position) emits exactly the tokens each sequence emits when run alone with the same uniforms."""
    from cosyvoice_amd.llm import Qwen2LM
    cfg = LlmConfig.tiny()
    sd = llm_state_dict(cfg, round_to=torch.bfloat16)
    lm = Qwen2LM(cfg, dtype=torch.bfloat16, max_batch=4, ctx_max=256, max_out=256).load_state_dict(sd)
    g = torch.Generator().manual_seed(21)
    shapes = [(5, 3, 7), (2, 0, 0), (9, 4, 12), (1, 1, 3)]          # (text, prompt text, prompt speech) lengths
    texts = [torch.randint(0, cfg.vocab_size, (1, a), generator=g, dtype=torch.int32) for a, _, _ in shapes]
    ptexts = [torch.randint(0, cfg.vocab_size, (1, b), generator=g, dtype=torch.int32) for _, b, _ in shapes]
    pspeech = [torch.randint(0, cfg.speech_token_size, (1, c), generator=g, dtype=torch.int32) for _, _, c in shapes]
    uni = torch.zeros(16, 101, 2)
    uni[:, :, 0] = torch.rand(16, 101, generator=g) * 0.6
    uni[:, :, 1] = torch.rand(16, 101, generator=g)
    got = lm.generate_batch(texts, ptexts, pspeech, uniforms=uni)
    assert len({len(t) for t in got}) > 1                                # sequences end at different steps
    for b, (a, _, _) in enumerate(shapes):
        assert 2 * a <= len(got[b]) <= 20 * a
        u1 = uni.clone()
        u1[0] = uni[b]
        alone = lm.generate_batch([texts[b]], [ptexts[b]], [pspeech[b]], uniforms=u1)[0]
        assert alone == got[b], (b, alone[:8], got[b][:8])


def test_generate_from_ready_made_prefill_embeddings():
    """lm_inputs hook: feeding the embedding sequence the standard assembly would build (llm.py:837-852) gives the same tokens
    as the id-based path, also in a ragged batch."""
    import torch.nn.functional as F
    from cosyvoice_amd.llm import Qwen2LM
    cfg = LlmConfig.tiny()
    sd = llm_state_dict(cfg, round_to=torch.bfloat16)
    lm = Qwen2LM(cfg, dtype=torch.bfloat16, max_batch=4, ctx_max=256, max_out=256).load_state_dict(sd)
    g = torch.Generator().manual_seed(33)
    shapes = [(4, 2, 5), (7, 0, 0)]
    texts = [torch.randint(0, cfg.vocab_size, (1, a), generator=g, dtype=torch.int32) for a, _, _ in shapes]
    ptexts = [torch.randint(0, cfg.vocab_size, (1, b), generator=g, dtype=torch.int32) for _, b, _ in shapes]
    pspeech = [torch.randint(0, cfg.speech_token_size, (1, c), generator=g, dtype=torch.int32) for _, _, c in shapes]
    uni = torch.zeros(16, 101, 2)
    uni[:, :, 0] = torch.rand(16, 101, generator=g) * 0.6
    uni[:, :, 1] = torch.rand(16, 101, generator=g)
    want = lm.generate_batch(texts, ptexts, pspeech, uniforms=uni)
    embs = []
    for t, pt, ps in zip(texts, ptexts, pspeech):
        te = F.embedding(torch.cat([pt, t], 1).long()[0], sd["llm.model.model.embed_tokens.weight"].float())
        pe = F.embedding(ps.long()[0], sd["speech_embedding.weight"].float())
        embs.append(torch.cat([sd["llm_embedding.weight"][0:1].float(), te, sd["llm_embedding.weight"][1:2].float(), pe], 0))
    got = lm.generate_batch(texts, ptexts, pspeech, uniforms=uni, lm_inputs=embs)
    assert got == want


@pytest.mark.parametrize("split_qkv", ["0", "1"])
def test_stage_abi_step_equals_python_composed_step(split_qkv, monkeypatch):
    """(both forms of the input RMSNorm: fused into the QKV kernel's prologue / its own cv_rmsnorm_reduce launch, the >8-row default)
    cv_llm_step_graph_create (the decode step composed + captured inside the library) against the same step composed launch by
    launch from Python: teacher-forced logits and free-running tokens (injected uniforms) bit-identical; the header's stage entry
    points are what a non-Python host would bind."""
    from cosyvoice_amd.llm import Qwen2LM
    cfg = LlmConfig.tiny()
    sd = llm_state_dict(cfg)
    g = torch.Generator().manual_seed(3)
    text = torch.randint(0, cfg.vocab_size, (1, 6), generator=g, dtype=torch.int32)
    ptext = torch.randint(0, cfg.vocab_size, (1, 3), generator=g, dtype=torch.int32)
    ps = torch.randint(0, cfg.speech_token_size, (1, 8), generator=g, dtype=torch.int32)
    forced = torch.randint(0, cfg.speech_token_size, (9,), generator=g).tolist()
    uni = torch.rand(16, 101, 2, generator=g) * 0.98
    monkeypatch.setenv("CV_SPLIT_QKV_NORM", split_qkv)
    outs = {}
    for abi in (True, False):
        lm = Qwen2LM(cfg, dtype=torch.bfloat16, max_batch=4, ctx_max=256, max_out=256)
        lm.use_stage_abi = abi
        lm.load_state_dict(sd)
        lp = lm.forced_logits(text, ptext, ps, forced).cpu()
        toks = lm.generate_batch([text, text], [ptext, ptext], [ps, ps[:, :5]], uniforms=uni, max_steps=40)
        outs[abi] = (lp, toks, len(lm._graphs))
    assert torch.equal(outs[True][0], outs[False][0])
    assert outs[True][1] == outs[False][1] and all(len(t) >= 10 for t in outs[True][1])
    assert outs[True][2] >= 1


def test_32_row_token_loop_equals_two_16_row_loops():
    """17..32 sequences per step run the skinny kernels with two 16-row MFMA groups per weight fragment (MR = 2: one weight stream
    for all rows).  Every row's dot products are formed in the same order as in the one-group form, so 20 sequences decoded in one
    loop must give exactly the tokens of the same sequences decoded as 10 + 10 (same injected uniforms per sequence)."""
    from cosyvoice_amd.llm import Qwen2LM
    cfg = LlmConfig.tiny()
    sd = llm_state_dict(cfg)
    g = torch.Generator().manual_seed(11)
    n = 20
    texts = [torch.randint(0, cfg.vocab_size, (1, 5 + b % 4), generator=g, dtype=torch.int32) for b in range(n)]
    ptext = torch.randint(0, cfg.vocab_size, (1, 3), generator=g, dtype=torch.int32)
    ps = [torch.randint(0, cfg.speech_token_size, (1, 6 + b % 3), generator=g, dtype=torch.int32) for b in range(n)]
    uni = torch.rand(32, 101, 2, generator=g) * 0.98
    lm32 = Qwen2LM(cfg, dtype=torch.bfloat16, max_batch=32, ctx_max=256, max_out=256).load_state_dict(sd)
    t32 = lm32.generate_batch(texts, [ptext] * n, ps, uniforms=uni, max_steps=40)
    lm16 = Qwen2LM(cfg, dtype=torch.bfloat16, max_batch=16, ctx_max=256, max_out=256).load_state_dict(sd)
    ta = lm16.generate_batch(texts[:10], [ptext] * 10, ps[:10], uniforms=uni[:16], max_steps=40)
    tb = lm16.generate_batch(texts[10:], [ptext] * 10, ps[10:], uniforms=uni[10:26], max_steps=40)
    assert all(len(t) >= 10 for t in t32)
    assert t32 == ta + tb
