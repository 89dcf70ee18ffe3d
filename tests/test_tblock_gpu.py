"""GPU: the fused row-block kernels of the estimator's BasicTransformerBlock (cv_tblock_head / cv_tblock_tail,
csrc/flowblock.hip) through the C ABI against a plain PyTorch fp32 restatement of the same ops
(/root/reference/cosyvoice/flow/components/transformer.py:243-316: pre-LN self-attention projections, to_out + residual,
LN, Linear + exact-erf GELU + Linear + residual), and the fused estimator against the unfused cv_gemm / cv_layernorm path.

Tolerances: operands (LN output, GELU output, weights) are 16-bit with fp32 accumulation; the fp32 reference rounds the same
intermediates to the operand type, so what remains is accumulation order and the erfc polynomial (|err| <= 1.5e-7)."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

C, INNER, FF = 256, 512, 1024


def _weights(dt, seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc)
    w = dict(g1=1 + 0.1 * r(C), b1=0.1 * r(C), wq=r(INNER, C, sc=C ** -0.5), wk=r(INNER, C, sc=C ** -0.5), wv=r(INNER, C, sc=C ** -0.5),
             wo=r(C, INNER, sc=INNER ** -0.5), bo=0.1 * r(C), g3=1 + 0.1 * r(C), b3=0.1 * r(C),
             w1=r(FF, C, sc=C ** -0.5), bf1=0.2 * r(FF), w2=r(C, FF, sc=FF ** -0.5), bf2=0.1 * r(C))
    for k in ("wq", "wk", "wv", "wo", "w1", "w2"):       # the values the kernels see
        w[k] = w[k].to(dt).float()
    return {k: v.cuda() for k, v in w.items()}


def _packed(w, dt):
    from cosyvoice_amd import ops
    return dict(wqkv_p=ops.pack_skinny(torch.cat([w["wq"], w["wk"], w["wv"]], 0).to(dt).contiguous()),
                wo_p=ops.pack_skinny(w["wo"].to(dt).contiguous()), w1_p=ops.pack_skinny(w["w1"].to(dt).contiguous()),
                w2_p=ops.pack_skinny(w["w2"].to(dt).contiguous()))


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("R,T", [(2, 150), (1, 64), (3, 1)])
@pytest.mark.parametrize("mt", ["4", "3", "2", "1"])
def test_head_vs_torch(dt, R, T, mt, monkeypatch):
    from cosyvoice_amd import ops
    monkeypatch.setenv("CV_TBLOCK_MT", mt)   # rows per workgroup = 16 MT
    w = _weights(dt)
    pk = _packed(w, dt)
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(R, T, C, generator=g) * 2 + 0.3).cuda()
    Tp = (T + 7) // 8 * 8
    qk = torch.zeros(R, T, 2 * INNER, device="cuda", dtype=dt)
    vt = torch.zeros(R, INNER // 64, 64, Tp, device="cuda", dtype=dt)
    p = ops.tblock_params(x, R, T, 1e-5, dt)
    p.g1, p.b1n, p.wqkv_p = w["g1"].data_ptr(), w["b1"].data_ptr(), pk["wqkv_p"].data_ptr()
    p.qk, p.ldqk, p.vt, p.vt_ld = qk.data_ptr(), 2 * INNER, vt.data_ptr(), Tp
    ops.tblock_head(p)
    torch.cuda.synchronize()
    xn = F.layer_norm(x, (C,), w["g1"], w["b1"], 1e-5).to(dt).float()
    ref_qk = torch.cat([xn @ w["wq"].t(), xn @ w["wk"].t()], -1)
    ref_v = xn @ w["wv"].t()                                         # (R, T, 512)
    tol = 2e-2 if dt == torch.bfloat16 else 3e-3
    e_qk = (qk.float() - ref_qk).abs().max().item()
    got_v = vt[..., :T].float().reshape(R, INNER, T).transpose(1, 2)
    e_v = (got_v - ref_v).abs().max().item()
    print(f"head[{dt},{R}x{T}]: qk Linf {e_qk:.3e}  vt Linf {e_v:.3e} (values ~ {ref_qk.abs().mean().item():.2f})")
    assert e_qk < tol and e_v < tol
    assert vt[..., T:].abs().max().item() == 0 if Tp > T else True   # the pad columns stay untouched


def _tail_ref(x, ao, w, dt):
    x1 = x if ao is None else x + ao.float() @ w["wo"].t() + w["bo"]
    xn = F.layer_norm(x1, (C,), w["g3"], w["b3"], 1e-5).to(dt).float()
    h = F.gelu(xn @ w["w1"].t() + w["bf1"]).to(dt).float()
    return x1 + h @ w["w2"].t() + w["bf2"]


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("R,T", [(2, 150), (1, 64), (2, 65)])
@pytest.mark.parametrize("outproj", [True, False])
@pytest.mark.parametrize("mt", ["4", "3", "2", "1"])
def test_tail_vs_torch(dt, R, T, outproj, mt, monkeypatch):
    from cosyvoice_amd import ops
    monkeypatch.setenv("CV_TBLOCK_MT", mt)
    w = _weights(dt, seed=2)
    pk = _packed(w, dt)
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(R, T, C, generator=g) * 2 + 0.3).cuda()
    ao = torch.randn(R, T, INNER, generator=g).cuda().to(dt)
    ref = _tail_ref(x, ao if outproj else None, w, dt)
    xk = x.clone()
    oa = torch.zeros(R, T, 2 * C, device="cuda", dtype=dt)
    p = ops.tblock_params(xk, R, T, 1e-5, dt)
    if outproj:
        p.ao, p.ldao, p.wo_p, p.bo = ao.data_ptr(), INNER, pk["wo_p"].data_ptr(), w["bo"].data_ptr()
    p.g3, p.b3n = w["g3"].data_ptr(), w["b3"].data_ptr()
    p.w1_p, p.bf1, p.w2_p, p.bf2 = pk["w1_p"].data_ptr(), w["bf1"].data_ptr(), pk["w2_p"].data_ptr(), w["bf2"].data_ptr()
    p.out_act, p.ldoa = oa[:, :, C:].data_ptr(), 2 * C
    ops.tblock_tail(p)
    torch.cuda.synchronize()
    tol = 3e-2 if dt == torch.bfloat16 else 4e-3
    e = (xk - ref).abs().max().item()
    e16 = (oa[:, :, C:].float() - ref).abs().max().item()
    print(f"tail[{dt},{R}x{T},outproj={outproj}]: Linf {e:.3e}  16-bit copy {e16:.3e} (values ~ {ref.abs().mean().item():.2f})")
    assert e < tol and e16 < tol + (6e-2 if dt == torch.bfloat16 else 8e-3)
    assert oa[:, :, :C].abs().max().item() == 0


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("R,T", [(2, 150), (1, 64), (2, 65), (3, 1)])
@pytest.mark.parametrize("mt", ["4", "3", "2", "1"])
def test_tail_head_fused_equals_tail_then_head(dt, R, T, mt, monkeypatch):
    """cv_tblock_tail_head (tail of block i + head of block i + 1 in one launch) against cv_tblock_tail followed by cv_tblock_head:
    the fp32 residual stream x must be bit-identical (same accumulators, stored once); [Q | K] and V^T may differ by the summation
    order of the LayerNorm statistics (taken from the accumulators instead of re-read rows) — a 16-bit ulp here and there — and both
    are checked against torch fp32."""
    from cosyvoice_amd import ops
    monkeypatch.setenv("CV_TBLOCK_MT", mt)
    w, w2 = _weights(dt, seed=2), _weights(dt, seed=7)       # block i, block i + 1
    pk, pk2 = _packed(w, dt), _packed(w2, dt)
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(R, T, C, generator=g) * 2 + 0.3).cuda()
    ao = torch.randn(R, T, INNER, generator=g).cuda().to(dt)
    Tp = (T + 7) // 8 * 8

    def run(fused):
        xk = x.clone()
        qk = torch.zeros(R, T, 2 * INNER, device="cuda", dtype=dt)
        vt = torch.zeros(R, INNER // 64, 64, Tp, device="cuda", dtype=dt)
        p = ops.tblock_params(xk, R, T, 1e-5, dt)
        p.ao, p.ldao, p.wo_p, p.bo = ao.data_ptr(), INNER, pk["wo_p"].data_ptr(), w["bo"].data_ptr()
        p.g3, p.b3n = w["g3"].data_ptr(), w["b3"].data_ptr()
        p.w1_p, p.bf1, p.w2_p, p.bf2 = pk["w1_p"].data_ptr(), w["bf1"].data_ptr(), pk["w2_p"].data_ptr(), w["bf2"].data_ptr()
        p.g1, p.b1n, p.wqkv_p = w2["g1"].data_ptr(), w2["b1"].data_ptr(), pk2["wqkv_p"].data_ptr()
        p.qk, p.ldqk, p.vt, p.vt_ld = qk.data_ptr(), 2 * INNER, vt.data_ptr(), Tp
        if fused:
            ops.tblock_tail_head(p)
        else:
            ops.tblock_tail(p)
            ops.tblock_head(p)
        torch.cuda.synchronize()
        return xk, qk, vt
    x_f, qk_f, vt_f = run(True)
    x_s, qk_s, vt_s = run(False)
    assert torch.equal(x_f, x_s)
    ref_x = _tail_ref(x, ao, w, dt)
    xn = F.layer_norm(x_s, (C,), w2["g1"], w2["b1"], 1e-5).to(dt).float()
    ref_qk = torch.cat([xn @ w2["wq"].t(), xn @ w2["wk"].t()], -1)
    ref_v = xn @ w2["wv"].t()
    tol = 2e-2 if dt == torch.bfloat16 else 3e-3
    got_v = vt_f[..., :T].float().reshape(R, INNER, T).transpose(1, 2)
    e_x, e_qk, e_v = (x_f - ref_x).abs().max().item(), (qk_f.float() - ref_qk).abs().max().item(), (got_v - ref_v).abs().max().item()
    d_qk, d_v = (qk_f.float() - qk_s.float()).abs().max().item(), (vt_f.float() - vt_s.float()).abs().max().item()
    print(f"tail+head[{dt},{R}x{T},MT={mt}]: x Linf {e_x:.3e}, qk {e_qk:.3e}, vt {e_v:.3e} vs torch; fused vs separate qk {d_qk:.2e} vt {d_v:.2e}")
    assert e_x < (3e-2 if dt == torch.bfloat16 else 4e-3) and e_qk < tol and e_v < tol
    assert d_qk < tol and d_v < tol
    assert vt_f[..., T:].abs().max().item() == 0 if Tp > T else True


def test_gelu_erf_polynomial():
    """The tail's erfc polynomial against torch's exact GELU over the whole useful range (through a 1-row FFN with identity-like
    weights is overkill: the tail test above already covers it at 16-bit; this pins the fp32 behaviour through W2 = I)."""
    from cosyvoice_amd import ops
    dt = torch.float16
    w = _weights(dt, seed=5)
    # LN off (gamma 1, beta 0 on a pre-normalised row is not expressible), so drive the hidden layer directly: W1 = [I; 0], bf1 = sweep
    w["w1"] = torch.zeros(FF, C).cuda()
    w["bf1"] = torch.linspace(-8, 8, FF).cuda()
    w["w2"] = torch.zeros(C, FF).cuda()
    w["w2"][0, :] = 1.0 / 64                     # column 0 of the output = sum of gelu(sweep) / 64
    w["bf2"] = torch.zeros(C).cuda()
    pk = _packed(w, dt)
    x = torch.randn(1, 64, C).cuda()
    xk = x.clone()
    p = ops.tblock_params(xk, 1, 64, 1e-5, dt)
    p.g3, p.b3n = w["g3"].data_ptr(), w["b3"].data_ptr()
    p.w1_p, p.bf1, p.w2_p, p.bf2 = pk["w1_p"].data_ptr(), w["bf1"].data_ptr(), pk["w2_p"].data_ptr(), w["bf2"].data_ptr()
    ops.tblock_tail(p)
    torch.cuda.synchronize()
    ref = (F.gelu(w["bf1"]).to(dt).float().sum() / 64).item()
    got = (xk - x)[0, :, 0]
    assert (got - ref).abs().max().item() < 2e-3 * max(1.0, abs(ref))


@pytest.mark.parametrize("dt,tol", [(torch.float16, 6e-3), (torch.bfloat16, 5e-2)])
def test_fused_estimator_equals_unfused(dt, tol, monkeypatch):
    """ConditionalDecoder.forward_cl with the row-block kernels vs the cv_gemm / cv_layernorm launches (CV_FLOW_FUSED=0) on the
    same weights and inputs, tiny depth, T not a multiple of 64, batch of 2 CFG pairs."""
    from cosyvoice_amd.config import FlowConfig
    from cosyvoice_amd.flow import CausalMaskedDiffWithXvec
    from cosyvoice_amd.weights import flow_state_dict
    cfg = FlowConfig.tiny()
    flow = CausalMaskedDiffWithXvec(cfg, dtype=dt).load_state_dict(flow_state_dict(cfg))
    est = flow.decoder.estimator
    assert est.fused
    R, T = 4, 150
    g = torch.Generator().manual_seed(0)
    xin = torch.randn(R, T, cfg.est_in_channels, generator=g).cuda().to(dt)
    tt = est.time_table([0.3])
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("CV_FLOW_FUSED", flag)
        ws = est._workspace(R, T)
        ws["xin"].copy_(xin)
        outs.append(est.forward_cl(ws, R, tt[0]).clone())
    torch.cuda.synchronize()
    d = (outs[0] - outs[1]).abs()
    print(f"fused vs unfused [{dt}]: Linf {d.max().item():.3e} L1 {d.mean().item():.3e} (values ~ {outs[1].abs().mean().item():.2f})")
    assert d.max().item() < tol * max(1.0, outs[1].abs().max().item())


@pytest.mark.parametrize("dt,tol", [(torch.float16, 6e-3), (torch.bfloat16, 5e-2)])
def test_estimator_tail_head_fusion_on_off(dt, tol, monkeypatch):
    """The estimator with cv_tblock_tail_head launches (default) against the three-launches-per-block composition, eager and through
    the stage ABI's captured solver: same kernels otherwise, LayerNorm statistics summed in a different order."""
    from cosyvoice_amd.config import FlowConfig
    from cosyvoice_amd.flow import CausalMaskedDiffWithXvec
    from cosyvoice_amd.weights import flow_state_dict
    cfg = FlowConfig.tiny()
    flow = CausalMaskedDiffWithXvec(cfg, dtype=dt).load_state_dict(flow_state_dict(cfg))
    est = flow.decoder.estimator
    R, T = 4, 150
    g = torch.Generator().manual_seed(0)
    xin = torch.randn(R, T, cfg.est_in_channels, generator=g).cuda().to(dt)
    tt = est.time_table([0.3])
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("CV_FLOW_FUSE_TAIL_HEAD", flag)
        assert est.fuse_tail_head == (flag == "1")
        ws = est._workspace(R, T)
        ws["xin"].copy_(xin)
        outs.append(est.forward_cl(ws, R, tt[0]).clone())
    torch.cuda.synchronize()
    d = (outs[0] - outs[1]).abs()
    print(f"estimator tail+head fusion on vs off [{dt}]: Linf {d.max().item():.3e} (values ~ {outs[1].abs().mean().item():.2f})")
    assert torch.isfinite(outs[0]).all() and d.max().item() < tol * max(1.0, outs[1].abs().max().item())
    # whole flow: eager composition == captured stage-ABI solver, with the fusion on
    monkeypatch.setenv("CV_FLOW_FUSE_TAIL_HEAD", "1")
    tok = torch.randint(0, cfg.vocab_size, (2, 14), generator=g, dtype=torch.int32)
    ptok = torch.randint(0, cfg.vocab_size, (2, 6), generator=g, dtype=torch.int32)
    pfeat = torch.clamp(torch.randn(2, 12, 80, generator=g) * 2 - 6, -11.5, 2.0)
    emb = torch.randn(2, cfg.spk_embed_dim, generator=g)
    flow.decoder.use_graph = False
    m_eager = flow.inference_batch(tok, ptok, pfeat, emb).clone()
    flow.decoder.use_graph = True
    flow.inference_batch(tok, ptok, pfeat, emb)
    m_graph = flow.inference_batch(tok, ptok, pfeat, emb).clone()
    assert torch.equal(m_eager, m_graph)


@pytest.mark.parametrize("dt,tol", [(torch.float16, 6e-3), (torch.bfloat16, 5e-2)])
@pytest.mark.parametrize("cin,T", [(320, 150), (256, 64), (512, 131)])
def test_resblock_vs_torch(dt, tol, cin, T):
    """cv_resblock_conv1 / conv2 (CausalResnetBlock1D: flow/decoder.py:36-56, components/decoder.py:54-59) against torch fp32 on the
    same 16-bit-rounded operands: causal k3 conv -> LayerNorm over channels -> Mish (+ time term), twice, + 1x1 conv of the input."""
    from cosyvoice_amd import _lib as L
    from cosyvoice_amd import ops
    g = torch.Generator().manual_seed(cin + T)
    R, Cc = 2, 256
    r = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    a = r(R, T, cin).to(dt)
    w1, w2, wr = r(Cc, cin, 3, sc=(3 * cin) ** -0.5).to(dt), r(Cc, Cc, 3, sc=(3 * Cc) ** -0.5).to(dt), r(Cc, cin, 1, sc=cin ** -0.5).to(dt)
    b1, b2, br = 0.1 * r(Cc), 0.1 * r(Cc), 0.1 * r(Cc)
    g1, be1, g2, be2 = 1 + 0.1 * r(Cc), 0.1 * r(Cc), 1 + 0.1 * r(Cc), 0.1 * r(Cc)
    tadd = 0.3 * r(Cc)

    def causal_conv(x, w, b):   # x (R, T, Ci) -> (R, T, Co), left padding k - 1
        k = w.shape[2]
        return F.conv1d(F.pad(x.float().transpose(1, 2), (k - 1, 0)), w.float(), b).transpose(1, 2)
    h = F.mish(F.layer_norm(causal_conv(a, w1, b1), (Cc,), g1, be1, 1e-5)) + tadd
    h16 = h.to(dt)
    ref = F.mish(F.layer_norm(causal_conv(h16, w2, b2), (Cc,), g2, be2, 1e-5)) + causal_conv(a, wr, br)

    def packed(w):              # (Co, Ci, k) -> (Co, k * Ci) with k = tap * Ci + ci, K zero-padded to a multiple of 128
        wm = w.permute(0, 2, 1).reshape(w.shape[0], -1)
        kp = (wm.shape[1] + 127) // 128 * 128
        wp = torch.zeros(w.shape[0], kp, dtype=dt)
        wp[:, :wm.shape[1]] = wm
        return ops.pack_skinny(wp.cuda().contiguous())
    dv = lambda t: t.cuda().contiguous()
    ad, h1d, outd = dv(a), torch.zeros(R, T, Cc, device="cuda", dtype=dt), torch.zeros(R, T, Cc, device="cuda")
    keep = [packed(w1), packed(w2), packed(wr)] + [dv(t) for t in (b1, g1, be1, tadd, b2, g2, be2, br)]
    p = L.ResblockParams()
    p.dtype, p.R, p.T, p.C, p.cin = L.TORCH_DT[dt], R, T, Cc, cin
    p.a, p.lda = ad.data_ptr(), cin
    p.w1_p, p.b1, p.g1, p.be1, p.tadd = keep[0].data_ptr(), keep[3].data_ptr(), keep[4].data_ptr(), keep[5].data_ptr(), keep[6].data_ptr()
    p.h1, p.ldh1 = h1d.data_ptr(), Cc
    p.w2_p, p.b2, p.g2, p.be2 = keep[1].data_ptr(), keep[7].data_ptr(), keep[8].data_ptr(), keep[9].data_ptr()
    p.wr_p, p.br = keep[2].data_ptr(), keep[10].data_ptr()
    p.out, p.ldo, p.eps = outd.data_ptr(), Cc, 1e-5
    for mt in ("4", "3", "2", "1"):
        os.environ["CV_TBLOCK_MT"] = mt
        h1d.zero_(); outd.zero_()
        ops._issue("cv_resblock_conv1", p)
        ops._issue("cv_resblock_conv2", p)
        torch.cuda.synchronize()
        e1 = (h1d.float().cpu() - h).abs().max().item()
        e2 = (outd.cpu() - ref).abs().max().item()
        print(f"resblock[{dt}, cin={cin}, T={T}, MT={mt}]: h1 Linf {e1:.3e}  out Linf {e2:.3e} (values ~ {ref.abs().mean().item():.2f})")
        assert e1 < tol and e2 < tol
    os.environ.pop("CV_TBLOCK_MT", None)


@pytest.mark.parametrize("ragged", [False, True])
def test_stage_abi_solver_equals_python_composed_solver(ragged, monkeypatch):
    """cv_flow_euler_graph_create (the library composes n_steps x [pack -> estimator -> Euler update] from a descriptor and captures
    it) vs the graph captured from cosyvoice_amd/flow.py's own launch sequence: same kernels, same order -> bit-identical state."""
    from cosyvoice_amd.config import FlowConfig
    from cosyvoice_amd.flow import CausalMaskedDiffWithXvec
    from cosyvoice_amd.weights import flow_state_dict
    cfg = FlowConfig.tiny()
    flow = CausalMaskedDiffWithXvec(cfg, dtype=torch.float16).load_state_dict(flow_state_dict(cfg))
    cfm = flow.decoder
    assert cfm.estimator.fused_all and cfm.use_stage_abi
    cfm.use_graph = True
    B, T = 2, 150
    g = torch.Generator().manual_seed(1)
    mu = torch.randn(B, T, 80, generator=g).cuda()
    cond = torch.randn(B, T, 80, generator=g).cuda()
    spks = torch.randn(B, 80, generator=g).cuda()
    klen = torch.tensor([T, T - 37], dtype=torch.int32, device="cuda") if ragged else None
    x0 = torch.randn(B, T, 80, generator=g).cuda()
    res = {}
    for abi in (True, False):
        cfm.use_stage_abi = abi
        x = x0.clone()
        cfm.solve(x, mu, spks, cond, 4, klen=klen)          # first call: eager warm-up + graph build
        x.copy_(x0)
        cfm.solve(x, mu, spks, cond, 4, klen=klen)          # second call: graph replay
        torch.cuda.synchronize()
        res[abi] = x.clone()
    assert len(cfm._graphs) == 2
    assert torch.isfinite(res[True]).all() and torch.equal(res[True], res[False])
