"""GPU: cv_gemm / cv_layernorm through the C ABI against plain torch fp32 references of the same op."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTS = [torch.float32, torch.bfloat16, torch.float16]
TOL = {torch.float32: 2e-5, torch.bfloat16: 2e-2, torch.float16: 3e-3}


def _rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-12)).item()


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("M,N,K", [(1000, 256, 256), (128, 1536, 256), (77, 80, 256), (2256, 4864, 896), (300, 18, 448)])
def test_linear_epilogues(dt, M, N, K):
    from cosyvoice_amd import ops
    torch.manual_seed(0)
    dev = "cuda"
    x = torch.randn(M, K, device=dev).to(dt)
    W = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt)
    bias = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev)
    ref = x.float() @ W.float().t() + bias + res
    o32 = torch.empty(M, N, device=dev)
    oa = torch.empty(M, N, device=dev, dtype=dt)
    ops.linear(x, W, bias=bias, res=res, act=ops.ACT_GELU, out_f32=o32, out_act=oa)
    torch.cuda.synchronize()
    assert _rel(o32, ref) < TOL[dt] * 0.5 + 1e-6
    assert _rel(oa, F.gelu(ref)) < TOL[dt]


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("Cin,Cout,k,dil,T", [(256, 256, 3, 1, 500), (64, 64, 11, 5, 3000), (320, 256, 3, 1, 333), (96, 512, 7, 1, 200)])
def test_conv1d_channels_last(dt, Cin, Cout, k, dil, T):
    from cosyvoice_amd import ops
    torch.manual_seed(1)
    dev = "cuda"
    B = 2
    x = torch.randn(B, T, Cin, device=dev).to(dt)
    w = (torch.randn(Cout, Cin, k, device=dev) / (Cin * k) ** 0.5).to(dt)
    bias = torch.randn(Cout, device=dev)
    alpha = torch.rand(Cout, device=dev) + 0.5
    pad = (k * dil - dil) // 2
    ref = F.conv1d(x.float().transpose(1, 2), w.float(), bias, dilation=dil, padding=pad).transpose(1, 2)
    Wp = w.permute(0, 2, 1).reshape(Cout, k * Cin).contiguous()
    o32 = torch.empty(B, T, Cout, device=dev)
    oa = torch.empty(B, T, Cout, device=dev, dtype=dt)
    ops.conv1d_cl(x, Wp, k, dilation=dil, pad_left=pad, bias=bias, act=ops.ACT_SNAKE, act_param=alpha, out_f32=o32, out_act=oa)
    torch.cuda.synchronize()
    assert _rel(o32, ref) < TOL[dt] * 0.5 + 1e-6
    sn = ref + (1.0 / (alpha + 1e-9)) * torch.sin(ref * alpha) ** 2
    assert _rel(oa, sn) < TOL[dt]
    # causal variant (left pad k-1), as CausalConv1d flow/decoder.py:59-85
    refc = F.conv1d(F.pad(x.float().transpose(1, 2), ((k - 1) * dil, 0)), w.float(), bias, dilation=dil).transpose(1, 2)
    ops.conv1d_cl(x, Wp, k, dilation=dil, pad_left=(k - 1) * dil, bias=bias, out_f32=o32)
    torch.cuda.synchronize()
    assert _rel(o32, refc) < TOL[dt] * 0.5 + 1e-6


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_qkv_split_mode(dt):
    from cosyvoice_amd import ops
    torch.manual_seed(2)
    dev = "cuda"
    B, T, C, H = 2, 200, 256, 8
    Tp = 208
    x = torch.randn(B, T, C, device=dev).to(dt)
    W = (torch.randn(3 * H * 64, C, device=dev) / C ** 0.5).to(dt)
    q = torch.zeros(B, T, H * 64, device=dev, dtype=dt)
    k = torch.zeros(B, T, H * 64, device=dev, dtype=dt)
    vt = torch.zeros(B, H, 64, Tp, device=dev, dtype=dt)
    ops.gemm(x, W, T, 3 * H * 64, C, batch=B, a_bs=(T * C, 0), lda=C, out_act=q, oa_bs=(T * H * 64, 0), ldoa=H * 64,
             qkv=dict(q_cols=H * 64, k_cols=H * 64, q_scale=0.125, k_out=k, k_bs=T * H * 64, ldk=H * 64, vt_out=vt,
                      vt_heads=H, vt_ld=Tp))
    torch.cuda.synchronize()
    ref = x.float() @ W.float().t()
    assert _rel(q, ref[..., :512] * 0.125) < TOL[dt]
    assert _rel(k, ref[..., 512:1024]) < TOL[dt]
    vref = ref[..., 1024:].reshape(B, T, H, 64).permute(0, 2, 3, 1)
    assert _rel(vt[..., :T], vref) < TOL[dt]


@pytest.mark.parametrize("odt", DTS)
@pytest.mark.parametrize("rows,dim,rms", [(1000, 256, False), (37, 512, False), (16, 896, True)])
def test_layernorm(odt, rows, dim, rms):
    from cosyvoice_amd import ops
    torch.manual_seed(3)
    dev = "cuda"
    x = torch.randn(rows, dim, device=dev) * 3 + 1
    g = torch.randn(dim, device=dev)
    b = None if rms else torch.randn(dim, device=dev)
    if rms:
        ref = g * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6))
    else:
        ref = F.layer_norm(x, (dim,), g, b, 1e-5)
    o32 = torch.empty(rows, dim, device=dev)
    oa = torch.empty(rows, dim, device=dev, dtype=odt)
    ops.layernorm(x, g, b, 1e-6 if rms else 1e-5, rms=rms, out_f32=o32, out_act=oa)
    torch.cuda.synchronize()
    assert (o32 - ref).abs().max().item() < 2e-5 * ref.abs().max().item() + 1e-5
    assert _rel(oa, ref) < {torch.float32: 1e-6, torch.bfloat16: 5e-3, torch.float16: 6e-4}[odt]


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_tall_gemm_tile_plain_and_swiglu(dt):
    """>= 8192 rows with N, K >= 512 take the 8-wave 128 x 128 tile (the LLM prefill of a 4-batch decode job): plain epilogue with a
    residual, and the SwiGLU epilogue over [gate16 | up16]-interleaved weights (the prefill's gate/up GEMM)."""
    from cosyvoice_amd import ops
    torch.manual_seed(1)
    dev, M, K, I = "cuda", 9024, 896, 1024
    x = (torch.randn(M, K, device=dev) * 0.5).to(dt)
    W = (torch.randn(896, K, device=dev) / K ** 0.5).to(dt)
    res = torch.randn(M, 896, device=dev)
    o32 = torch.empty(M, 896, device=dev)
    ops.linear(x, W, res=res, out_f32=o32)
    g = (torch.randn(I, K, device=dev) / K ** 0.5).to(dt)
    u = (torch.randn(I, K, device=dev) / K ** 0.5).to(dt)
    gu = torch.stack([g.view(I // 16, 16, K), u.view(I // 16, 16, K)], dim=1).reshape(2 * I, K).contiguous()
    h = torch.empty(M, I, device=dev, dtype=dt)
    ops.linear(x, gu, act=ops.ACT_SWIGLU, out_act=h)
    torch.cuda.synchronize()
    assert _rel(o32, x.float() @ W.float().t() + res) < TOL[dt] * 0.5
    assert _rel(h, F.silu(x.float() @ g.float().t()) * (x.float() @ u.float().t())) < TOL[dt]


def _unsplit(t):
    """pre-split fp32-sized storage (.., C) -> the fp32 values hi + lo it stands for."""
    raw = t.contiguous().view(torch.int16).reshape(*t.shape[:-1], t.shape[-1] // 8, 16)
    hi, lo = raw[..., :8].contiguous().view(torch.bfloat16), raw[..., 8:].contiguous().view(torch.bfloat16)
    return (hi.float() + lo.float()).reshape(t.shape)


@pytest.mark.parametrize("shape", ["auto", "4x1", "2x2", "2x1", "1x2", "1x1"])
@pytest.mark.parametrize("Cin,Cout,k,dil,T,B", [(64, 64, 3, 1, 1000, 2), (64, 64, 11, 5, 777, 1), (128, 128, 7, 3, 601, 2),
                                                (256, 256, 11, 5, 500, 1), (256, 256, 3, 1, 130, 3), (128, 64, 7, 1, 70, 2), (64, 192, 3, 3, 257, 1)])
def test_conv_window_kernel_presplit(Cin, Cout, k, dil, T, B, shape, monkeypatch):
    """conv_win_kernel (stride-1 bf16x3 convs on pre-split operands: activation window in LDS, weights streamed L2 -> VGPR) against torch
    fp32, and against gemm_kernel on the same launch (the default; the window kernel is opt-in, CV_CONV_WIN=1): both the fp32 output and the pre-split snake output, every
    workgroup shape, ragged row / column tails, zero padding on both sides, residual + scale epilogue."""
    from cosyvoice_amd import ops, _lib as L
    from cosyvoice_amd.hift import _presplit
    torch.manual_seed(11)
    dev = "cuda"
    x = torch.randn(B, T, Cin)
    w = torch.randn(Cout, Cin, k) / (Cin * k) ** 0.5
    bias = torch.randn(Cout, device=dev)
    alpha = torch.rand(Cout, device=dev) + 0.5
    res = torch.randn(B, T, Cout, device=dev)
    pad = (k * dil - dil) // 2
    xs = _presplit(x.reshape(B * T, Cin)).view(torch.float32).reshape(B, T, Cin).to(dev)
    Wp = w.permute(0, 2, 1).reshape(Cout, k * Cin).contiguous()
    Ws = _presplit(Wp).view(torch.float32).to(dev)
    xv, wv = _unsplit(xs.cpu()).to(dev), _unsplit(Ws.cpu()).to(dev).reshape(Cout, k, Cin).permute(0, 2, 1)
    ref = (F.conv1d(xv.double().transpose(1, 2), wv.double(), bias.double(), dilation=dil, padding=pad).transpose(1, 2) + res.double()) * 0.5
    sn = ref + (1.0 / (alpha.double() + 1e-9)) * torch.sin(ref * alpha.double()) ** 2
    outs = {}
    for win in ("1", "0"):
        monkeypatch.setenv("CV_CONV_WIN", win)
        if shape != "auto":
            monkeypatch.setenv("CV_CONV_WIN_SHAPE", shape)
        o32 = torch.full((B, T, Cout), float("nan"), device=dev)
        oa = torch.full((B, T, Cout), float("nan"), device=dev)
        ops.conv1d_cl(xs, Ws, k, dilation=dil, pad_left=pad, bias=bias, res=res, out_scale=0.5, act=ops.ACT_SNAKE, act_param=alpha,
                      out_f32=o32, out_act=oa, dtype=L.CV_F32X3, x3_flags=7)
        torch.cuda.synchronize()
        outs[win] = (o32.clone(), _unsplit(oa.cpu()).to(dev))
        # bf16x3 drops the lo*lo term: relative 2^-16 per product, fp32 accumulation
        assert (o32.double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
        assert (outs[win][1].double() - sn).abs().max().item() < 5e-5 * max(1.0, sn.abs().max().item())
    d = (outs["1"][0] - outs["0"][0]).abs().max().item()
    assert d < 2e-6 * max(1.0, ref.abs().max().item())
    if Cin == 64:   # one channel chunk: same summation order as the K-tile loop
        assert torch.equal(outs["1"][0], outs["0"][0]) and torch.equal(outs["1"][1], outs["0"][1])
