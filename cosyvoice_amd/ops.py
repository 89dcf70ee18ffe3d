"""Thin tensor-level wrappers over the C ABI (device pointers out of torch tensors; torch is plumbing only)."""
import atexit
import ctypes as C
import sys
import threading
import weakref

import torch

from . import _lib as L
from ._lib import (ACT_ELU, ACT_GELU, ACT_LEAKY, ACT_MISH, ACT_NONE, ACT_SILU, ACT_SNAKE, ACT_SWIGLU, ACT_TANH,  # noqa: F401
                   CV_BF16, CV_F16, CV_F32, OUT_QKV, OUT_ROWMAJOR)


def _req_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("cosyvoice_amd ops need device tensors (no CPU path)")


_tls = threading.local()


class f32_products:
    """``with ops.f32_products("bf16x3"):`` — fp32 GEMMs / convs issued by this thread inside the block compute their
    products as three bf16 MFMAs on hi/lo splits (cv_dtype CV_F32X3: ~2^-16 relative, fp32 accumulate, ~3x the exact-f32
    MFMA rate) instead of the exact-f32 MFMA.  "exact" restores the default."""

    def __init__(self, mode: str):
        assert mode in ("exact", "bf16x3")
        self.mode = mode

    def __enter__(self):
        self.prev = getattr(_tls, "f32_mode", "exact")
        _tls.f32_mode = self.mode

    def __exit__(self, *exc):
        _tls.f32_mode = self.prev


SHAPE_CACHE_CAP = 64


def bound_cache(cache: dict, key, *dependents, cap=None):
    """Workspaces, projected position tables and captured graphs are cached per request shape; a service sees an open-ended set
    of lengths.  Call before inserting ``key``: when ``cache`` already holds ``cap`` other shapes, the device is drained and
    ``cache`` plus every dict in ``dependents`` (caches holding pointers into it: captured graphs, buffers) are emptied — a rare
    full flush instead of unbounded growth.  Returns True when a flush happened."""
    cap = SHAPE_CACHE_CAP if cap is None else cap
    if key in cache or len(cache) < cap:
        return False
    torch.cuda.synchronize()
    cache.clear()
    for d in dependents:
        d.clear()
    return True


def drop_graphs(*caches):
    """Destroy every captured graph held in the given per-shape caches (values that are ops.Graph objects) and empty them, after
    draining the device.  Called when the tensors a graph / descriptor has raw pointers into are about to be replaced
    (load_state_dict on a model that has already run)."""
    if not any(caches):
        return
    torch.cuda.synchronize()
    for c in caches:
        for g in c.values():
            if hasattr(g, "destroy"):
                g.destroy()
        c.clear()


class Recorder:
    """Collects the parameter blocks of cv_gemm / cv_layernorm / cv_attention calls instead of launching them, so a fixed
    launch sequence (the v1 LM's cached decode step) is built once and re-issued with a few patched fields per step: the
    Python cost of a launch drops from filling a 50-field struct to one foreign call."""

    def __init__(self):
        self.calls = []          # [(entry point name, params struct)]

    def __enter__(self):
        self._prev = getattr(_tls, "recorder", None)
        _tls.recorder = self
        return self

    def __exit__(self, *exc):
        _tls.recorder = self._prev

    def add_raw(self, name, args):
        """A launch with positional arguments (a mutable list the owner may patch; the stream is appended at replay)."""
        self.calls.append((name, args))

    def replay(self):
        lib, st = L.lib(), L.stream_ptr()
        for name, p in self.calls:
            rc = getattr(lib, name)(*p, st) if isinstance(p, list) else getattr(lib, name)(C.byref(p), st)
            if rc != 0:
                raise RuntimeError(f"{name} failed with cv_status {rc}")


def _issue(name, p):
    rec = getattr(_tls, "recorder", None)
    if rec is not None:
        rec.calls.append((name, p))
        return
    L.check(getattr(L.lib(), name)(C.byref(p), L.stream_ptr()), name)


def gemm(A, W, M, N, K, *, dtype=None, batch=1, batch_inner=0, a_bs=(0, 0), lda=None, a_rows=0, cin=0,
         a_row_stride=1, tap_base=0, tap_step=0, w_bs=(0, 0), ldw=None, bias=None, res=None, res_bs=(0, 0), ldres=0,
         res2=None, ldres2=0, out_scale=1.0, act=ACT_NONE, act_param=None, act_slope=0.0, out_f32=None, o32_bs=(0, 0),
         ldo32=0, out_act=None, oa_bs=(0, 0), ldoa=0, out_row_stride=1, out_row_off=0, out_rows=0, qkv=None, x3_flags=0):
    _req_cuda(A, W, bias, res, res2, out_f32, out_act)
    p = L.GemmParams()
    p.dtype = L.TORCH_DT[A.dtype] if dtype is None else dtype
    if dtype is None and A.dtype == torch.float32 and getattr(_tls, "f32_mode", "exact") == "bf16x3":
        p.dtype = L.CV_F32X3
    p.M, p.N, p.K, p.batch, p.batch_inner = M, N, K, batch, batch_inner
    p.A, p.a_bs0, p.a_bs1, p.lda, p.a_rows = A.data_ptr(), a_bs[0], a_bs[1], (lda if lda is not None else A.stride(-2)), a_rows
    p.cin, p.a_row_stride, p.tap_base, p.tap_step = cin, a_row_stride, tap_base, tap_step
    p.W, p.w_bs0, p.w_bs1, p.ldw = W.data_ptr(), w_bs[0], w_bs[1], (ldw if ldw is not None else W.stride(-2))
    p.bias = L.ptr(bias)
    p.res, p.res_bs0, p.res_bs1, p.ldres = L.ptr(res), res_bs[0], res_bs[1], ldres
    p.res2, p.ldres2 = L.ptr(res2), ldres2
    p.out_scale, p.act, p.act_param, p.act_slope = out_scale, act, L.ptr(act_param), act_slope
    p.out_f32, p.o32_bs0, p.o32_bs1, p.ldo32 = L.ptr(out_f32), o32_bs[0], o32_bs[1], ldo32
    p.out_act, p.oa_bs0, p.oa_bs1, p.ldoa = L.ptr(out_act), oa_bs[0], oa_bs[1], ldoa
    p.out_row_stride, p.out_row_off, p.out_rows = out_row_stride, out_row_off, out_rows
    p.x3_flags = x3_flags if p.dtype == L.CV_F32X3 else 0   # pre-split operand / output storage of the bf16x3 path (cv_gemm_params.x3_flags)
    if qkv is not None:
        p.out_mode = OUT_QKV
        p.q_cols, p.k_cols, p.q_scale = qkv["q_cols"], qkv["k_cols"], qkv.get("q_scale", 1.0)
        p.k_out, p.k_bs, p.ldk = qkv["k_out"].data_ptr(), qkv["k_bs"], qkv["ldk"]
        p.vt_out, p.vt_heads, p.vt_ld = qkv["vt_out"].data_ptr(), qkv["vt_heads"], qkv["vt_ld"]
    _issue("cv_gemm", p)


def linear(x, W, *, bias=None, res=None, act=ACT_NONE, act_param=None, act_slope=0.0, out_f32=None, out_act=None,
           out_scale=1.0):
    """x (rows, K) contiguous rows (stride(0) = ld); W (N, K)."""
    M, K = x.shape
    N = W.shape[0]
    gemm(x, W, M, N, K, bias=bias, res=res, ldres=(res.stride(0) if res is not None else 0), act=act,
         act_param=act_param, act_slope=act_slope, out_f32=out_f32, ldo32=(out_f32.stride(0) if out_f32 is not None else 0),
         out_act=out_act, ldoa=(out_act.stride(0) if out_act is not None else 0), out_scale=out_scale)


def conv1d_cl(x, Wp, taps, *, dilation=1, pad_left=0, stride=1, T_out=None, **kw):
    """Channels-last conv: x (B, T, Cin) -> (B, T_out, Cout); Wp (Cout, taps*Cin) with k = tap*Cin + ci.
    kw: bias/res/act/... as gemm, with out tensors shaped (B, T_out, Cout)."""
    B, T, Cin = x.shape
    N = Wp.shape[0]
    if T_out is None:
        T_out = T
    out_f32, out_act, res, res2 = kw.pop("out_f32", None), kw.pop("out_act", None), kw.pop("res", None), kw.pop("res2", None)
    gemm(x, Wp, T_out, N, taps * Cin, batch=B, a_bs=(x.stride(0), 0), lda=x.stride(1), a_rows=T, cin=Cin,
         a_row_stride=stride, tap_base=-pad_left, tap_step=dilation,
         res=res, res_bs=((res.stride(0), 0) if res is not None else (0, 0)), ldres=(res.stride(1) if res is not None else 0),
         res2=res2, ldres2=(res2.stride(1) if res2 is not None else 0),
         out_f32=out_f32, o32_bs=((out_f32.stride(0), 0) if out_f32 is not None else (0, 0)),
         ldo32=(out_f32.stride(1) if out_f32 is not None else 0),
         out_act=out_act, oa_bs=((out_act.stride(0), 0) if out_act is not None else (0, 0)),
         ldoa=(out_act.stride(1) if out_act is not None else 0), **kw)


def layernorm(x, gamma, beta, eps, *, rms=False, add=None, rows_per_group=0, act=ACT_NONE, out_f32=None, out_act=None,
              out_scale=1.0):
    """x (rows, dim) fp32."""
    _req_cuda(x, gamma, beta, add, out_f32, out_act)
    p = L.NormParams()
    p.rows, p.dim, p.rms, p.eps = x.shape[0], x.shape[1], int(rms), eps
    p.x, p.ldx = x.data_ptr(), x.stride(0)
    p.gamma, p.beta = L.ptr(gamma), L.ptr(beta)
    p.add, p.add_ld, p.rows_per_group = L.ptr(add), (add.stride(0) if add is not None else 0), rows_per_group
    p.act, p.out_scale = act, out_scale
    p.out_dtype = L.TORCH_DT[out_act.dtype] if out_act is not None else CV_F32
    p.out_f32, p.ldo32 = L.ptr(out_f32), (out_f32.stride(0) if out_f32 is not None else 0)
    p.out_act, p.ldoa = L.ptr(out_act), (out_act.stride(0) if out_act is not None else 0)
    _issue("cv_layernorm", p)


def attention(q, k, vt, out, *, H, Hkv, Tq, Tk, scale, q_bs, ldq, k_bs, ldk, vt_ld, o_bs, ldo, B, klen=None, chunk=0,
              causal=False, causal_off=0, bias=None, bias_bs=0, bias_hs=0, bias_ld=0, q_hs=0, k_hs=0, q_off=0):
    _req_cuda(q, k, vt, out, klen, bias)
    p = L.AttnParams()
    p.dtype = L.TORCH_DT[q.dtype]
    p.B, p.H, p.Hkv, p.Tq, p.Tk = B, H, Hkv, Tq, Tk
    p.q, p.q_bs, p.ldq = q.data_ptr(), q_bs, ldq
    p.k, p.k_bs, p.ldk = k.data_ptr(), k_bs, ldk
    p.vt, p.vt_ld = vt.data_ptr(), vt_ld
    p.out, p.o_bs, p.ldo = out.data_ptr(), o_bs, ldo
    p.scale, p.klen, p.chunk, p.causal, p.causal_off = scale, L.ptr(klen), chunk, int(causal), causal_off
    p.bias, p.bias_bs, p.bias_hs, p.bias_ld = L.ptr(bias), bias_bs, bias_hs, bias_ld
    p.q_hs, p.k_hs, p.q_off = q_hs, k_hs, q_off
    _issue("cv_attention", p)


# ----------------------------------------------------------------------------- fused transformer block (CFM estimator)
def tblock_params(x, R, T, eps, dtype):
    """x (R, T, 256) fp32 residual stream -> parameter block shared by tblock_head / tblock_tail (fill the stage fields)."""
    _req_cuda(x)
    p = L.TBlockParams()
    p.dtype, p.R, p.T = L.TORCH_DT[dtype], R, T
    p.C, p.inner, p.ff = 256, 512, 1024
    p.x, p.ldx, p.eps = x.data_ptr(), x.stride(-2), eps
    return p


def tblock_head(p):
    _issue("cv_tblock_head", p)


def tblock_tail(p):
    _issue("cv_tblock_tail", p)


def tblock_tail_head(p):
    """Tail of block i (tail fields of p) + head of block i + 1 (head fields of p) in one launch."""
    _issue("cv_tblock_tail_head", p)


# ----------------------------------------------------------------------------- layout / HiFT helpers
def to_channels_last(x, out):
    """x (B,C,T) fp32 -> out (B,T,ld) any dtype, zero-filled pad columns."""
    _req_cuda(x, out)
    B, Cc, T = x.shape
    L.check(L.lib().cv_to_channels_last(C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), L.TORCH_DT[out.dtype], B, Cc, T,
                                        out.stride(1), L.stream_ptr()), "cv_to_channels_last")


def to_channels_first(x, out, Cc=None):
    """x (B,T,ld) fp32 -> out (B,C,T) fp32."""
    _req_cuda(x, out)
    B, T, _ = x.shape
    Cc = out.shape[1] if Cc is None else Cc
    L.check(L.lib().cv_to_channels_first(C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), B, Cc, T, x.stride(1),
                                         L.stream_ptr()), "cv_to_channels_first")


def snake_multi(x2d, alphas, outs, split=False):
    """x2d (rows, C) fp32; alphas: list of (C,) fp32; outs: list of (rows, C) tensors of one dtype.  ``split`` (fp32 outs): write the
    pre-split chunk format of the bf16x3 convs ([4 x bf16 hi | 4 x bf16 lo] per 4 values)."""
    _req_cuda(x2d, *alphas, *outs)
    n = len(alphas)
    a = (C.c_void_p * n)(*[t.data_ptr() for t in alphas])
    o = (C.c_void_p * n)(*[t.data_ptr() for t in outs])
    L.check(L.lib().cv_snake_multi(C.c_void_p(x2d.data_ptr()), x2d.shape[0], x2d.shape[1], x2d.stride(0), n, a, o,
                                   outs[0].stride(0), L.CV_F32X3 if split else L.TORCH_DT[outs[0].dtype], L.stream_ptr()), "cv_snake_multi")


def stft16(s, out):
    """s (B,S) fp32 -> out (B, S/4+1, ld>=18)."""
    _req_cuda(s, out)
    L.check(L.lib().cv_stft16(C.c_void_p(s.data_ptr()), C.c_void_p(out.data_ptr()), L.TORCH_DT[out.dtype], s.shape[0], s.shape[1],
                              out.stride(1), L.stream_ptr()), "cv_stft16")


def istft16(y, wav, audio_limit):
    """y (B,F,ld>=18) fp32 -> wav (B,(F-1)*4) fp32."""
    _req_cuda(y, wav)
    L.check(L.lib().cv_istft16(C.c_void_p(y.data_ptr()), C.c_void_p(wav.data_ptr()), y.shape[0], y.shape[1], y.stride(1),
                               C.c_float(audio_limit), L.stream_ptr()), "cv_istft16")


def hift_source(f0, phase_vec, noise, lin_w, lin_b, work, s, up, sampling_rate, sine_amp, noise_std, vthr):
    _req_cuda(f0, phase_vec, noise, lin_w, lin_b, work, s)
    B, T = f0.shape
    nh = phase_vec.shape[1]
    L.check(L.lib().cv_hift_source(C.c_void_p(f0.data_ptr()), C.c_void_p(phase_vec.data_ptr()), C.c_void_p(noise.data_ptr()),
                                   C.c_void_p(lin_w.data_ptr()), C.c_void_p(lin_b.data_ptr()), C.c_void_p(work.data_ptr()),
                                   C.c_void_p(s.data_ptr()), B, T, up, nh, C.c_float(sampling_rate), C.c_float(sine_amp),
                                   C.c_float(noise_std), C.c_float(vthr), L.stream_ptr()), "cv_hift_source")


# ----------------------------------------------------------------------------- flow helpers / graphs
def embedding(table, idx, out):
    """table (V,dim) fp32, idx (rows,) int32, out (rows, ld) any dtype."""
    _req_cuda(table, idx, out)
    L.check(L.lib().cv_embedding(C.c_void_p(table.data_ptr()), C.c_void_p(idx.data_ptr()), C.c_void_p(out.data_ptr()),
                                 L.TORCH_DT[out.dtype], idx.numel(), table.shape[1], out.stride(-2), L.stream_ptr()), "cv_embedding")


def est_pack(x, mu, spks, cond, xin):
    _req_cuda(x, mu, spks, cond, xin)
    B, T, Cc = x.shape
    L.check(L.lib().cv_est_pack(C.c_void_p(x.data_ptr()), C.c_void_p(mu.data_ptr()), C.c_void_p(spks.data_ptr()),
                                C.c_void_p(cond.data_ptr()), C.c_void_p(xin.data_ptr()), L.TORCH_DT[xin.dtype], B, T, Cc,
                                L.stream_ptr()), "cv_est_pack")


def cfm_update(x, v, dt, cfg_rate):
    _req_cuda(x, v)
    B, T, Cc = x.shape
    L.check(L.lib().cv_cfm_update(C.c_void_p(x.data_ptr()), C.c_void_p(v.data_ptr()), B, T, Cc, C.c_float(dt), C.c_float(cfg_rate),
                                  L.stream_ptr()), "cv_cfm_update")


def groupnorm_workspace(B, T, groups, device):
    f = L.lib().cv_groupnorm_workspace_floats
    f.restype = C.c_int64
    return torch.empty(int(f(B, T, groups)), device=device, dtype=torch.float32)


def groupnorm_cl(x, groups, gamma, beta, eps, partial, *, act=ACT_NONE, add=None, out_f32=None, out_act=None):
    """x (B, T, C) fp32 channels-last (any row / batch strides); add (C,) or (B, C) fp32 added after the activation."""
    _req_cuda(x, gamma, beta, partial, add, out_f32, out_act)
    B, T, Cc = x.shape
    assert x.stride(2) == 1 and partial.numel() >= B * groups * ((T + 31) // 32) * 2
    p = L.GroupNormParams()
    p.B, p.T, p.C, p.groups, p.eps = B, T, Cc, groups, eps
    p.x, p.x_bs, p.ldx = x.data_ptr(), x.stride(0), x.stride(1)
    p.gamma, p.beta = L.ptr(gamma), L.ptr(beta)
    p.add, p.add_ld = L.ptr(add), (add.stride(0) if add is not None and add.dim() == 2 else 0)
    p.act = act
    p.out_dtype = L.TORCH_DT[out_act.dtype] if out_act is not None else CV_F32
    if out_f32 is not None:
        p.out_f32, p.o32_bs, p.ldo32 = out_f32.data_ptr(), out_f32.stride(0), out_f32.stride(1)
    if out_act is not None:
        p.out_act, p.oa_bs, p.ldoa = out_act.data_ptr(), out_act.stride(0), out_act.stride(1)
    p.partial = partial.data_ptr()
    L.check(L.lib().cv_groupnorm_cl(C.byref(p), L.stream_ptr()), "cv_groupnorm_cl")


def interp_linear_cl(x, y):
    """x (T_in, C) fp32 rows (stride(0) = ld) -> y (T_out, C) any dtype: F.interpolate(mode='linear') along T."""
    _req_cuda(x, y)
    assert x.dtype == torch.float32 and x.shape[1] == y.shape[1] and x.stride(1) == 1 and y.stride(1) == 1
    L.check(L.lib().cv_interp_linear_cl(C.c_void_p(x.data_ptr()), x.stride(0), x.shape[0], C.c_void_p(y.data_ptr()), y.stride(0),
                                        L.TORCH_DT[y.dtype], y.shape[0], x.shape[1], L.stream_ptr()), "cv_interp_linear_cl")


class Graph:
    """hipGraph of a launch sequence issued through the ABI on torch's current stream.  Every live graph is registered so that
    ``ops.close()`` (also an atexit hook) can destroy it while the HIP runtime is still alive."""

    def __init__(self):
        self.handle = C.c_void_p(None)
        _GRAPHS.add(self)

    def capture(self, fn):
        # capture needs a non-default stream; replays may go to any stream
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            st = L.stream_ptr()
            L.check(L.lib().cv_graph_begin(st), "cv_graph_begin")
            try:
                fn()
            finally:
                rc = L.lib().cv_graph_end(st, C.byref(self.handle))
            L.check(rc, "cv_graph_end")
        torch.cuda.current_stream().wait_stream(side)
        return self

    @classmethod
    def from_llm_step(cls, desc):
        """Graph of one Qwen2 decode step, composed and captured by the library itself (cv_llm_step_graph_create)."""
        self = cls()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            L.check(L.lib().cv_llm_step_graph_create(C.byref(desc), L.stream_ptr(), C.byref(self.handle)), "cv_llm_step_graph_create")
        torch.cuda.current_stream().wait_stream(side)
        self._desc = desc   # keeps the host-side layer array alive
        return self

    @classmethod
    def from_flow_solver(cls, desc, keep=()):
        """Graph of a whole CFM Euler solve (n_steps x [pack -> estimator -> update]), composed and captured by the library
        (cv_flow_euler_graph_create).  ``keep``: host-side arrays the descriptor points to."""
        self = cls()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            L.check(L.lib().cv_flow_euler_graph_create(C.byref(desc), L.stream_ptr(), C.byref(self.handle)), "cv_flow_euler_graph_create")
        torch.cuda.current_stream().wait_stream(side)
        self._desc, self._keep = desc, keep
        return self

    def launch(self):
        """Replay on torch's current stream: as a hipGraphExec, or — when that stream is CU-masked (``masked_stream``) —
        launch by launch, because hipGraph replays ignore a stream's CU mask."""
        if torch.cuda.current_stream().cuda_stream in _MASKED_STREAMS:
            L.check(L.lib().cv_graph_launch_direct(self.handle, L.stream_ptr()), "cv_graph_launch_direct")
        else:
            L.check(L.lib().cv_graph_launch(self.handle, L.stream_ptr()), "cv_graph_launch")

    def launch_direct(self):
        L.check(L.lib().cv_graph_launch_direct(self.handle, L.stream_ptr()), "cv_graph_launch_direct")

    @property
    def num_launches(self):
        return L.lib().cv_graph_num_launches(self.handle)

    def destroy(self):
        """Release the hipGraph / hipGraphExec now (idempotent).  The owner must have drained the streams it was launched on."""
        h, self.handle = self.handle, C.c_void_p(None)
        if h:
            L.lib().cv_graph_destroy(h)

    def __del__(self):
        # objects collected after close() / at interpreter shutdown hold no handle any more (close() ran from atexit while the
        # runtime was alive); never call into HIP from a finalising interpreter
        try:
            if self.handle and not sys.is_finalizing():
                self.destroy()
        except Exception:
            pass


_GRAPHS = weakref.WeakSet()
_MASKED_STREAMS = {}  # raw handle -> ExternalStream of the CU-masked streams made here: Graph.launch issues direct launches on them


def destroy_masked_stream(stream):
    """hipStreamDestroy of a stream made by ``masked_stream`` (after draining it) and removal from the registry, so that a
    recycled handle value is never mistaken for a masked stream."""
    h = stream.cuda_stream
    if _MASKED_STREAMS.pop(h, None) is not None:
        stream.synchronize()
        L.check(L.lib().cv_stream_destroy(C.c_void_p(h)), "cv_stream_destroy")


def close():
    """Drain the device and release everything this module created through the ABI that the HIP runtime owns: captured graphs
    and CU-masked streams.  Registered with atexit (after torch's own hooks, so it runs before them): objects owned by the
    runtime that outlive it were what made profiled runs abort inside __cxa_finalize (round-1 records)."""
    if not _GRAPHS and not _MASKED_STREAMS:
        return
    try:
        if torch.cuda.is_available():
            torch.cuda.synchronize()
    except Exception:
        pass
    for g in list(_GRAPHS):
        try:
            g.destroy()
        except Exception:
            pass
    for h, st in list(_MASKED_STREAMS.items()):
        try:
            L.lib().cv_stream_destroy(C.c_void_p(h))
        except Exception:
            pass
    _MASKED_STREAMS.clear()


atexit.register(close)


def masked_stream(keep, n_xcd=8, slots=32, device=None):
    """Stream restricted to the CUs for which ``keep(slot, xcd)`` is true (slot 0..slots-1 within XCD 0..n_xcd-1).
    Mask bit i = (xcd i % n_xcd, slot i // n_xcd) — KFD's symmetric CU-mask mapping, confirmed on MI355X by timing.
    Every XCD must keep at least one CU (an XCD with an empty mask falls back to all of its CUs)."""
    n = n_xcd * slots
    words = (C.c_uint32 * ((n + 31) // 32))()
    for i in range(n):
        if keep(i // n_xcd, i % n_xcd):
            words[i // 32] |= 1 << (i % 32)
    for x in range(n_xcd):
        if not any(keep(s, x) for s in range(slots)):
            raise ValueError(f"CU mask leaves XCD {x} empty")
    st = C.c_void_p()
    L.check(L.lib().cv_stream_create_cumask(words, len(words), C.byref(st)), "cv_stream_create_cumask")
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    ext = torch.cuda.ExternalStream(st.value, device=dev)
    _MASKED_STREAMS[st.value] = ext
    return ext


# ----------------------------------------------------------------------------- LLM decode helpers
def pack_skinny(W, interleave=False):
    """W (N,K) 16-bit device -> packed MFMA B-fragment stream (N padded to 16)."""
    _req_cuda(W)
    N, K = W.shape
    ntiles = (N + 15) // 16
    Wp = torch.empty(ntiles * (K // 32) * 64 * 8, device=W.device, dtype=W.dtype)
    L.check(L.lib().cv_pack_skinny(C.c_void_p(W.data_ptr()), C.c_void_p(Wp.data_ptr()), N, K, int(interleave), L.stream_ptr()),
            "cv_pack_skinny")
    return Wp


def skinny_gemm(A, Wp, M, N, K, *, bias=None, ksplit=1, mode=0, out_f32=None, ldo=0, slab_stride=0, out_act=None, ldoa=0,
                norm=None, max_wgs=0, split_out=None, split_in=None):
    """norm = dict(x=, gamma=, eps=, x_out=None, slabs=None, nslab=0, slab_stride=0, ld_slab=0): fused RMSNorm prologue
    (A is then unused; pass any 16-bit tensor for the dtype)."""
    _req_cuda(A, Wp, bias, out_f32, out_act)
    p = L.SkinnyParams()
    p.dtype, p.M, p.N, p.K = L.TORCH_DT[A.dtype], M, N, K
    p.A, p.lda, p.Wp, p.bias = A.data_ptr(), A.stride(0), Wp.data_ptr(), L.ptr(bias)
    p.ksplit, p.mode = ksplit, mode
    p.out_f32, p.ldo, p.slab_stride = L.ptr(out_f32), ldo, slab_stride
    p.out_act, p.ldoa = L.ptr(out_act), ldoa
    if norm is not None:
        x = norm["x"]
        _req_cuda(x, norm["gamma"], norm.get("x_out"), norm.get("slabs"))
        p.nx, p.ldnx = x.data_ptr(), x.stride(0)
        p.nslabs, p.n_nslab = L.ptr(norm.get("slabs")), norm.get("nslab", 0)
        p.nslab_stride, p.ld_nslab = norm.get("slab_stride", 0), norm.get("ld_slab", 0)
        p.ngamma, p.neps = norm["gamma"].data_ptr(), norm["eps"]
        p.nx_out = L.ptr(norm.get("x_out"))
    p.max_wgs = max_wgs
    if split_out is not None:   # producer of a split RMSNorm (mode 1): dict(xb=(16, N) 16-bit rows, ss=(N / 16, 16) fp32 partials)
        _req_cuda(split_out["xb"], split_out["ss"])
        p.xb_out, p.ldxb, p.ss_part = split_out["xb"].data_ptr(), split_out["xb"].stride(0), split_out["ss"].data_ptr()
    if split_in is not None:    # consumer (mode 2): dict(rs=partials, n=number of partial rows, eps=)
        _req_cuda(split_in["rs"])
        p.rs_part, p.n_rs_part, p.rs_eps = split_in["rs"].data_ptr(), split_in["n"], split_in["eps"]
    _issue("cv_skinny_gemm", p)


def rmsnorm_reduce(x, gamma, eps, xn, rows, *, slabs=None, nslab=0, slab_stride=0, ld_slab=0):
    _req_cuda(x, gamma, xn, slabs)
    L.check(L.lib().cv_rmsnorm_reduce(C.c_void_p(x.data_ptr()), x.stride(0), C.c_void_p(L.ptr(slabs)), nslab, C.c_int64(slab_stride), ld_slab,
                                      C.c_void_p(gamma.data_ptr()), C.c_float(eps), C.c_void_p(xn.data_ptr()), xn.stride(0),
                                      L.TORCH_DT[xn.dtype], rows, x.shape[1], L.stream_ptr()), "cv_rmsnorm_reduce")


def rope_append(qkv, pos_base, rows, rows_per_seq, Hq, Hkv, inv_freq, q_out, kcache, vtcache, ctx_max):
    _req_cuda(qkv, pos_base, inv_freq, q_out, kcache, vtcache)
    L.check(L.lib().cv_rope_append(C.c_void_p(qkv.data_ptr()), qkv.stride(0), C.c_void_p(pos_base.data_ptr()), rows, rows_per_seq,
                                   Hq, Hkv, C.c_void_p(inv_freq.data_ptr()), C.c_void_p(q_out.data_ptr()), q_out.stride(0),
                                   C.c_void_p(kcache.data_ptr()), C.c_void_p(vtcache.data_ptr()), ctx_max, L.TORCH_DT[q_out.dtype],
                                   L.stream_ptr()), "cv_rope_append")


def kv_retile(k_rm, vt_rm, k_tiled, vt_tiled, B, Hkv, ctx_max, n_keys):
    """Row-major caches (rope_append's layout) -> the fragment-tiled caches the fused decode_attention reads and appends to."""
    _req_cuda(k_rm, vt_rm, k_tiled, vt_tiled)
    L.check(L.lib().cv_kv_retile(C.c_void_p(k_rm.data_ptr()), C.c_void_p(vt_rm.data_ptr()), C.c_void_p(k_tiled.data_ptr()),
                                 C.c_void_p(vt_tiled.data_ptr()), B, Hkv, ctx_max, n_keys, L.stream_ptr()), "cv_kv_retile")


def kv_tile_index(ctx_max, key, d, v=False):
    """Element offset of (key, d) inside one (sequence, kv head) fragment-tiled cache (include/cosyvoice_amd.h, cv_kv_retile)."""
    t, r = key >> 6, key & 63
    if not v:
        i = ((((r >> 4) * 2 + (d >> 5)) * 64 + (r & 15) + 16 * ((d >> 3) & 3)) << 3) + (d & 7)
    else:
        r32 = r & 31
        i = ((((d >> 4) * 2 + (r >> 5)) * 64 + (d & 15) + 16 * ((r32 & 15) >> 2)) << 3) + (r32 & 3) + 4 * (r32 >> 4)
    return t * 4096 + i


def decode_attention(q, kcache, vtcache, ctx_len, ctx_add, out, B, Hq, Hkv, ctx_max, scale, qkv=None, inv_freq=None):
    """qkv (fp32 [B][ld]) + inv_freq: fused RoPE + KV append + attention (q then only carries the dtype)."""
    _req_cuda(q, kcache, vtcache, ctx_len, out, qkv, inv_freq)
    L.check(L.lib().cv_decode_attention(C.c_void_p(q.data_ptr()), q.stride(0), C.c_void_p(kcache.data_ptr()),
                                        C.c_void_p(vtcache.data_ptr()), C.c_void_p(ctx_len.data_ptr()), ctx_add,
                                        C.c_void_p(out.data_ptr()), out.stride(0), B, Hq, Hkv, ctx_max, C.c_float(scale),
                                        L.TORCH_DT[q.dtype], C.c_void_p(L.ptr(qkv)), (qkv.stride(0) if qkv is not None else 0),
                                        C.c_void_p(L.ptr(inv_freq)), L.stream_ptr()), "cv_decode_attention")


def sample_ras(params):
    L.check(L.lib().cv_sample_ras(C.byref(params), L.stream_ptr()), "cv_sample_ras")
