"""HiFT vocoder on MI355X — host side of the drop-in for the reference's ``HiFTGenerator``
(/root/reference/cosyvoice/hifigan/generator.py:223-411) and ``ConvRNNF0Predictor``
(hifigan/f0_predictor.py:19-55).  Same constructor hyper-parameters (via HiftConfig), same
``load_state_dict`` key names (legacy weight-norm ``weight_g/weight_v`` folded at load,
SURVEY.md §8b (i)), same ``inference(speech_feat, cache_source) -> (wav (B,S), source (B,1,S))``
and ``decode(x, s)`` signatures.  All arithmetic runs in libcosyvoice_amd.so (HIP, gfx950):
channels-last activations, every Conv1d / ConvTranspose1d as an implicit-GEMM MFMA launch with
the Snake / leaky-relu / residual / 1/3-mean fused into the epilogues.
"""
import math
import os
from typing import Dict, Optional

import torch

from . import ops
from .config import HiftConfig
from .weights import fold_weight_norm, hift_downsample_plan


def _get_padding(k, d=1):
    return int((k * d - d) / 2)


def _round_up(x, m):
    return (x + m - 1) // m * m


def _presplit(w32: torch.Tensor) -> torch.Tensor:
    """fp32 (rows, K) with K % 8 == 0 -> the pre-split storage of the bf16x3 GEMM (cv_gemm_params.x3_flags): every 8 values
    become [8 x bf16 hi | 8 x bf16 lo] in the same 32 bytes, hi = bf16(w), lo = bf16(w - hi).  Returned as int16 (rows, 2K)."""
    w32 = w32.to(torch.float32).cpu()
    hi = w32.to(torch.bfloat16)
    lo = (w32 - hi.to(torch.float32)).to(torch.bfloat16)
    rows, K = w32.shape
    assert K % 8 == 0
    out = torch.cat([hi.view(rows, K // 8, 8), lo.view(rows, K // 8, 8)], dim=2)   # (rows, K/8, 16) bf16
    return out.contiguous().view(torch.int16).reshape(rows, 2 * K)


class _Conv:
    """Packed Conv1d: W' [Cout][tap*Cin_pad + ci]."""

    def __init__(self, w, b, dtype, device, dilation=1, pad_left=0, stride=1):
        cout, cin, k = w.shape
        ch = 4 if dtype == torch.float32 else 8
        self.cin_pad = _round_up(cin, ch)
        wp = torch.zeros(cout, k, self.cin_pad, dtype=torch.float32)
        wp[:, :, :cin] = w.permute(0, 2, 1)
        self.w = wp.reshape(cout, k * self.cin_pad).to(device=device, dtype=dtype).contiguous()
        self.w_s = _presplit(wp.reshape(cout, k * self.cin_pad)).to(device) if (dtype == torch.float32 and self.cin_pad % 8 == 0) else None
        self.b = b.to(device=device, dtype=torch.float32).contiguous() if b is not None else None
        self.k, self.cout, self.dilation, self.pad_left, self.stride = k, cout, dilation, pad_left, stride


class _ConvT:
    """ConvTranspose1d(stride u, kernel k, padding p) as u phase GEMMs:
    out[q*u + r] = sum_m x[q + c_r - m] . w[:, :, j0_r + m*u],  j0_r = (r+p) % u, c_r = (r+p) // u."""

    def __init__(self, w, b, u, p, dtype, device):
        cin, cout, k = w.shape
        self.u, self.cout, self.cin = u, cout, cin
        self.b = b.to(device=device, dtype=torch.float32).contiguous()
        self.phases = []
        for r in range(u):
            j0, c = (r + p) % u, (r + p) // u
            taps = list(range(j0, k, u))
            wp = torch.stack([w[:, :, j].t() for j in taps], dim=1)  # (cout, ntaps, cin)
            w2 = wp.reshape(cout, len(taps) * cin)
            self.phases.append((w2.to(device=device, dtype=dtype).contiguous(), len(taps), c,
                                _presplit(w2).to(device) if (dtype == torch.float32 and cin % 8 == 0) else None))


class HiFTGenerator:
    def __init__(self, cfg: Optional[HiftConfig] = None, dtype: torch.dtype = torch.float32, device: str = "cuda",
                 f32_products: str = "bf16x3"):
        self.cfg = cfg or HiftConfig.v2()
        self.dtype = dtype
        # fp32 operands (the reference never halves HiFT): "exact" = f32 MFMA; "bf16x3" = three bf16 MFMAs on hi/lo splits per
        # product (~2^-16 relative, ~3x faster) for the decoder convs.  The F0 predictor always runs exact: its output feeds a
        # phase accumulation over 240 000 samples.
        self.f32_products = f32_products
        self.device = torch.device(device)
        self.num_kernels = len(self.cfg.resblock_kernel_sizes)
        self.num_upsamples = len(self.cfg.upsample_rates)
        self._loaded = False
        self._ws: Dict[tuple, dict] = {}
        self.use_stage_abi = os.environ.get("CV_HIFT_STAGE_ABI", "1") != "0"   # decode composed by cv_hift_decode_enqueue

    @property
    def presplit(self):
        """fp32 tensors with bf16x3 products: activations between the decoder's convs and the weights are kept PRE-SPLIT (hi / lo bf16
        planes per 4 values, cv_gemm_params.x3_flags) — the producer's epilogue splits once what every consuming conv would otherwise
        split once per tap and tile column.  Bit-identical to the plain bf16x3 launches (CV_HIFT_PRESPLIT=0), tested."""
        chans = [self.cfg.base_channels // 2 ** i for i in range(self.num_upsamples + 1)]   # 8-value groups along the channel axis
        return (self.dtype == torch.float32 and self.f32_products == "bf16x3" and all(c % 8 == 0 for c in chans)
                and os.environ.get("CV_HIFT_PRESPLIT", "1") != "0")

    # -- torch.nn.Module-like surface used by cli/model.py:72-81
    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    def load_state_dict(self, sd, strict: bool = False):
        # per-shape workspaces cache the decode descriptor (raw pointers of every conv / phase / alpha tensor): a reload replaces
        # those tensors, so everything derived from the old ones goes first
        if self._ws:
            torch.cuda.synchronize()
            self._ws.clear()
        sd = {k.replace("generator.", ""): v.detach().to("cpu", torch.float32) for k, v in sd.items()}  # model.py:79
        cfg, dt, dev = self.cfg, self.dtype, self.device
        f32 = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()

        def conv(name, **kw):
            return _Conv(fold_weight_norm(sd, name), sd[f"{name}.bias"], dt, dev, **kw)

        def resblock(name, k, dils):
            rb = []
            for j, d in enumerate(dils):
                rb.append(dict(c1=conv(f"{name}.convs1.{j}", dilation=d, pad_left=_get_padding(k, d)),
                               c2=conv(f"{name}.convs2.{j}", dilation=1, pad_left=_get_padding(k, 1)),
                               a1=f32(sd[f"{name}.activations1.{j}.alpha"]), a2=f32(sd[f"{name}.activations2.{j}.alpha"])))
            return rb

        self.conv_pre = conv("conv_pre", pad_left=3)
        self.ups = [_ConvT(fold_weight_norm(sd, f"ups.{i}"), sd[f"ups.{i}.bias"], u, (k - u) // 2, dt, dev)
                    for i, (u, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes))]
        self.source_downs, self.source_resblocks = [], []
        for i, (stride, ks, pad) in enumerate(hift_downsample_plan(cfg)):
            self.source_downs.append(_Conv(sd[f"source_downs.{i}.weight"], sd[f"source_downs.{i}.bias"], dt, dev,
                                           pad_left=pad, stride=stride))
            self.source_resblocks.append(resblock(f"source_resblocks.{i}", cfg.source_resblock_kernel_sizes[i],
                                                  cfg.source_resblock_dilation_sizes[i]))
        self.resblocks = []
        for i in range(self.num_upsamples):
            for j in range(self.num_kernels):
                self.resblocks.append(resblock(f"resblocks.{i * self.num_kernels + j}", cfg.resblock_kernel_sizes[j],
                                               cfg.resblock_dilation_sizes[j]))
        self.conv_post = conv("conv_post", pad_left=3)
        self.f0_convs = [conv(f"f0_predictor.condnet.{idx}", pad_left=1) for idx in (0, 2, 4, 6, 8)]
        ch = 4 if dt == torch.float32 else 8
        fc = cfg.f0_cond_channels
        wcls = torch.zeros(4, fc)  # N padded to 4 so the vector epilogue applies; row 0 is the classifier
        wcls[0] = sd["f0_predictor.classifier.weight"][0]
        self.f0_cls_w = wcls.to(device=dev, dtype=dt).contiguous()
        bcls = torch.zeros(4)
        bcls[0] = sd["f0_predictor.classifier.bias"][0]
        self.f0_cls_b = f32(bcls)
        self.lin_w = f32(sd["m_source.l_linear.weight"].reshape(-1))
        self.lin_b = f32(sd["m_source.l_linear.bias"].reshape(-1))
        self._stft_ld = _round_up(cfg.n_fft + 2, ch)
        self._loaded = True
        return self

    # ------------------------------------------------------------------ workspaces (no allocation in steady state)
    def _workspace(self, B, T):
        key = (B, T)
        ws = self._ws.get(key)
        if ws is not None:
            return ws
        ops.bound_cache(self._ws, key)
        cfg, dt, dev = self.cfg, self.dtype, self.device
        e = lambda *shape, dtype=torch.float32: torch.empty(*shape, device=dev, dtype=dtype)
        ws = {}
        S = T * cfg.total_upsample
        F = S // cfg.hop_len + 1
        ws["mel_cl"] = e(B, T, self.conv_pre.cin_pad, dtype=dt)
        ws["stft"] = e(B, F, self._stft_ld, dtype=dt)
        ws["a_pre"] = e(B, T, cfg.base_channels, dtype=dt)
        lens, chans = [], []
        t = T
        for i, u in enumerate(cfg.upsample_rates):
            t = t * u + (1 if i == self.num_upsamples - 1 else 0)
            lens.append(t)
            chans.append(cfg.base_channels // 2 ** (i + 1))
        ws["lens"], ws["chans"] = lens, chans
        for i, (t, c) in enumerate(zip(lens, chans)):
            ws[f"x{i}"] = e(B, t, c)           # stage input (ups + source), fp32 residual stream
            ws[f"xa{i}"] = [e(B, t, c, dtype=dt) for _ in range(self.num_kernels)]  # snake'd copies per resblock
            ws[f"r{i}"] = [e(B, t, c), e(B, t, c)]  # resblock residual ping-pong
            ws[f"ta{i}"] = e(B, t, c, dtype=dt)   # conv1 -> snake -> conv2 intermediate
            ws[f"ra{i}"] = e(B, t, c, dtype=dt)   # activated residual for the next conv1
            ws[f"acc{i}"] = [e(B, t, c), e(B, t, c)]  # running sum over the parallel resblocks
            ws[f"si{i}"] = [e(B, t, c), e(B, t, c)]   # source branch
            ws[f"out{i}"] = e(B, t, c, dtype=dt)      # leaky-relu'd stage output feeding the next layer
        ws["post"] = e(B, lens[-1], self._stft_ld)
        ws["wav"] = e(B, (lens[-1] - 1) * cfg.hop_len)
        # f0 predictor / source
        fc = cfg.f0_cond_channels
        ws["f0_a"] = [e(B, T, fc, dtype=dt), e(B, T, fc, dtype=dt)]
        ws["f0_y"] = e(B, T, 4)
        ws["f0"] = e(B, T)
        ws["src_work"] = torch.empty(B, cfg.nb_harmonics + 1, T, device=dev, dtype=torch.float64)
        ws["s"] = e(B, S)
        self._ws[key] = ws
        return ws

    # ------------------------------------------------------------------ building blocks
    def _conv(self, c: _Conv, x, T_out=None, flags=0, **kw):
        """``flags`` = cv_gemm_params.x3_flags (decoder convs under ``presplit``): 3 = operands pre-split (the packed weights are then
        c.w_s), 4 = write out_act pre-split."""
        w = c.w_s.view(torch.float32) if (flags & 2) else c.w
        if kw.get("out_act") is None:
            flags &= ~4
        ops.conv1d_cl(x, w, c.k, dilation=c.dilation, pad_left=c.pad_left, stride=c.stride, T_out=T_out, bias=c.b, x3_flags=flags, **kw)

    def _resblock(self, rb, x32, xa, ws, i, final):
        """x32: fp32 block input; xa: snake_{a1[0]}(x32) in dtype.  ``final`` = kwargs of the last conv2's epilogue
        (out_f32 / out_act / res2 / out_scale / act ...).  generator.py:91-98."""
        r, ta, ra = ws[f"r{i}"], ws[f"ta{i}"], ws[f"ra{i}"]
        cur32, cur_a = x32, xa
        n = len(rb)
        fl = 7 if self.presplit else 0
        for j, blk in enumerate(rb):
            self._conv(blk["c1"], cur_a, flags=fl, act=ops.ACT_SNAKE, act_param=blk["a2"], out_act=ta)
            if j < n - 1:
                nxt = r[j & 1]
                self._conv(blk["c2"], ta, flags=fl, res=cur32, out_f32=nxt, act=ops.ACT_SNAKE, act_param=rb[j + 1]["a1"], out_act=ra)
                cur32, cur_a = nxt, ra
            else:
                self._conv(blk["c2"], ta, flags=fl, res=cur32, **final)

    # ------------------------------------------------------------------ public API
    @torch.no_grad()
    def f0_predictor(self, speech_feat: torch.Tensor) -> torch.Tensor:
        """speech_feat (B,80,T) fp32 -> f0 (B,T).  f0_predictor.py:52-55."""
        assert self._loaded
        B, _, T = speech_feat.shape
        ws = self._workspace(B, T)
        x = speech_feat.to(self.device, torch.float32).contiguous()
        ops.to_channels_last(x, ws["mel_cl"])
        self._f0_from_cl(ws, B, T)
        return ws["f0"]

    def _f0_from_cl(self, ws, B, T):
        cur = ws["mel_cl"]
        for n, c in enumerate(self.f0_convs):
            out = ws["f0_a"][n & 1]
            self._conv(c, cur, act=ops.ACT_ELU, out_act=out)
            cur = out
        fc = self.cfg.f0_cond_channels
        ops.gemm(cur, self.f0_cls_w, B * T, 4, fc, lda=fc, bias=self.f0_cls_b, out_f32=ws["f0_y"], ldo32=4)
        torch.abs(ws["f0_y"][:, :, 0], out=ws["f0"])  # |.| of one column: data-movement-class torch op

    @torch.no_grad()
    def decode(self, x: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
        """x (B,80,T) mel, s (B,1,T*up) source -> wav (B, T*up).  generator.py:349-381."""
        assert self._loaded
        B, _, T = x.shape
        ws = self._workspace(B, T)
        ops.to_channels_last(x.to(self.device, torch.float32).contiguous(), ws["mel_cl"])
        s2 = s.to(self.device, torch.float32).reshape(B, -1).contiguous()
        return self._decode_cl(ws, B, T, s2)

    def _decode_cl(self, ws, B, T, s2):
        if self.use_stage_abi:
            # stage-level ABI: the library composes the whole decode from a descriptor (cv_hift_decode_enqueue: one foreign call
            # instead of ~100); _decode_cl_impl stays as the cross-check of the C composition (tests/test_hift_gpu.py)
            from . import _lib as L
            import ctypes as C
            key = ("_desc", self.presplit)
            ent = ws.get(key)
            if ent is None:
                ent = ws[key] = self.decode_desc(ws, B, T)
            desc = ent[0]
            desc.gemm_dtype = L.CV_F32X3 if (self.dtype == torch.float32 and self.f32_products == "bf16x3") else L.TORCH_DT[self.dtype]
            desc.s = s2.data_ptr()
            ops._req_cuda(s2)
            L.check(L.lib().cv_hift_decode_enqueue(C.byref(desc), L.stream_ptr()), "cv_hift_decode_enqueue")
            return ws["wav"]
        with ops.f32_products(self.f32_products):
            return self._decode_cl_impl(ws, B, T, s2)

    def decode_desc(self, ws, B, T):
        """cv_hift_decode_desc over this workspace (include/cosyvoice_amd.h) + the host arrays it points to."""
        from . import _lib as L
        import ctypes as C
        cfg = self.cfg
        keep = []

        ps = self.presplit
        fl_in, fl = (4 if ps else 0), (7 if ps else 0)

        def conv(c, cin, flags=None):
            flags = fl if flags is None else flags
            h = L.HiftConv()
            h.w, h.b = (c.w_s if (flags & 2) else c.w).data_ptr(), (c.b.data_ptr() if c.b is not None else None)
            h.k, h.cin, h.cout, h.dilation, h.pad_left, h.stride = c.k, cin, c.cout, c.dilation, c.pad_left, c.stride
            h.x3_flags = flags
            return h

        def resblock(rb, ch):
            units = (L.HiftResunit * len(rb))()
            keep.append(units)
            for j, blk in enumerate(rb):
                units[j].c1, units[j].c2 = conv(blk["c1"], ch), conv(blk["c2"], ch)
                units[j].a1, units[j].a2 = blk["a1"].data_ptr(), blk["a2"].data_ptr()
            r = L.HiftResblock()
            r.units, r.n_units = units, len(rb)
            return r

        d = L.HiftDecodeDesc()
        d.dtype, d.B, d.T, d.S = L.TORCH_DT[self.dtype], B, T, T * cfg.total_upsample
        d.n_stages, d.n_kernels, d.stft_ld, d.hop = self.num_upsamples, self.num_kernels, self._stft_ld, cfg.hop_len
        d.lrelu_slope, d.audio_limit = cfg.lrelu_slope, cfg.audio_limit
        d.presplit = int(ps)
        d.conv_pre = conv(self.conv_pre, ws["mel_cl"].shape[2], fl_in)
        stages = (L.HiftStage * self.num_upsamples)()
        keep.append(stages)
        nk = self.num_kernels
        for i in range(self.num_upsamples):
            g, up = stages[i], self.ups[i]
            t_out, c = ws["lens"][i], ws["chans"][i]
            ph = (L.HiftPhase * up.u)()
            keep.append(ph)
            for r, (wp, ntaps, cr, wps) in enumerate(up.phases):
                ph[r].w, ph[r].ntaps, ph[r].tap_base = (wps if ps else wp).data_ptr(), ntaps, cr
            g.phases, g.up_b, g.u, g.up_cin, g.up_flags = ph, up.b.data_ptr(), up.u, up.cin, (3 if ps else 0)
            g.source_down = conv(self.source_downs[i], self._stft_ld, fl_in)
            g.source_rb = resblock(self.source_resblocks[i], c)
            rbs = (L.HiftResblock * nk)(*[resblock(rb, c) for rb in self.resblocks[i * nk:(i + 1) * nk]])
            xa = (C.c_void_p * nk)(*[t.data_ptr() for t in ws[f"xa{i}"]])
            keep += [rbs, xa]
            g.rbs, g.xa, g.t_out, g.c = rbs, xa, t_out, c
            g.x32, g.r0, g.r1 = ws[f"x{i}"].data_ptr(), ws[f"r{i}"][0].data_ptr(), ws[f"r{i}"][1].data_ptr()
            g.ta, g.ra = ws[f"ta{i}"].data_ptr(), ws[f"ra{i}"].data_ptr()
            g.acc0, g.acc1 = ws[f"acc{i}"][0].data_ptr(), ws[f"acc{i}"][1].data_ptr()
            g.si0, g.si1, g.out = ws[f"si{i}"][0].data_ptr(), ws[f"si{i}"][1].data_ptr(), ws[f"out{i}"].data_ptr()
        d.stages = stages
        d.conv_post = conv(self.conv_post, ws["chans"][-1])
        d.mel_cl, d.stft, d.a_pre = ws["mel_cl"].data_ptr(), ws["stft"].data_ptr(), ws["a_pre"].data_ptr()
        d.post, d.wav = ws["post"].data_ptr(), ws["wav"].data_ptr()
        return d, keep

    def _decode_cl_impl(self, ws, B, T, s2):
        cfg = self.cfg
        ops.stft16(s2, ws["stft"])
        ps = self.presplit
        fl_in, fl = (4 if ps else 0), (7 if ps else 0)   # first convs read plain fp32 (mel, STFT) and write pre-split activations
        self._conv(self.conv_pre, ws["mel_cl"], flags=fl_in, act=ops.ACT_LEAKY, act_slope=cfg.lrelu_slope, out_act=ws["a_pre"])
        cur_a = ws["a_pre"]
        t_in = T
        nk = self.num_kernels
        for i in range(self.num_upsamples):
            t_out, c = ws["lens"][i], ws["chans"][i]
            last = i == self.num_upsamples - 1
            # source branch: strided conv of the source STFT, then its ResBlock (generator.py:361-363)
            sd_, srb = self.source_downs[i], self.source_resblocks[i]
            si0, si1 = ws[f"si{i}"]
            xa_src = ws[f"xa{i}"][0]  # free here: the main resblocks have not started
            self._conv(sd_, ws["stft"], T_out=t_out, flags=fl_in, out_f32=si0, act=ops.ACT_SNAKE, act_param=srb[0]["a1"], out_act=xa_src)
            self._resblock(srb, si0, xa_src, ws, i, dict(out_f32=si1))
            # ups[i] (+ reflect pad on the last stage) + source fusion: x = ups(x) + si  (generator.py:355-364)
            up = self.ups[i]
            x32 = ws[f"x{i}"]
            off = 1 if last else 0
            upf = 3 if ps else 0
            for r, (wp, ntaps, cr, wps) in enumerate(up.phases):
                ops.gemm(cur_a, wps.view(torch.float32) if ps else wp, t_in, c, ntaps * up.cin, batch=B, a_bs=(cur_a.stride(0), 0), lda=cur_a.stride(1),
                         a_rows=t_in, cin=up.cin, tap_base=cr, tap_step=-1, bias=up.b, res=si1,
                         res_bs=(si1.stride(0), 0), ldres=c, out_f32=x32, o32_bs=(x32.stride(0), 0), ldo32=c,
                         out_row_stride=up.u, out_row_off=r + off, out_rows=t_out, x3_flags=upf, ldw=ntaps * up.cin)
            if last:
                # ReflectionPad1d((1,0)): padded[0] = ups_out[1] = phase r=1, q=0
                wp, ntaps, cr, wps = up.phases[1]
                ops.gemm(cur_a, wps.view(torch.float32) if ps else wp, 1, c, ntaps * up.cin, batch=B, a_bs=(cur_a.stride(0), 0), lda=cur_a.stride(1),
                         a_rows=t_in, cin=up.cin, tap_base=cr, tap_step=-1, bias=up.b, res=si1,
                         res_bs=(si1.stride(0), 0), ldres=c, out_f32=x32, o32_bs=(x32.stride(0), 0), ldo32=c,
                         out_row_stride=up.u, out_row_off=0, out_rows=t_out, x3_flags=upf, ldw=ntaps * up.cin)
            # parallel ResBlocks, mean over kernels (generator.py:366-372)
            rbs = self.resblocks[i * nk:(i + 1) * nk]
            xas = ws[f"xa{i}"]
            ops.snake_multi(x32.view(B * t_out, c), [rb[0]["a1"] for rb in rbs], [a.view(B * t_out, c) for a in xas], split=ps)
            acc = ws[f"acc{i}"]
            slope = 0.01 if last else cfg.lrelu_slope  # F.leaky_relu default after the loop (generator.py:374)
            for j, rb in enumerate(rbs):
                if j < nk - 1:
                    fin = dict(out_f32=acc[j & 1])
                    if j > 0:
                        fin["res2"] = acc[(j - 1) & 1]
                else:
                    fin = dict(out_scale=1.0 / nk, act=ops.ACT_LEAKY, act_slope=slope, out_act=ws[f"out{i}"])
                    if j > 0:
                        fin["res2"] = acc[(j - 1) & 1]
                self._resblock(rb, x32, xas[j], ws, i, fin)
            cur_a = ws[f"out{i}"]
            t_in = t_out
        self._conv(self.conv_post, cur_a, flags=fl, out_f32=ws["post"])
        ops.istft16(ws["post"], ws["wav"], cfg.audio_limit)
        return ws["wav"]

    @torch.no_grad()
    def source(self, f0: torch.Tensor, phase_vec: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None):
        """f0 (B,T) -> s (B, T*up).  Randoms are drawn on the device when not injected (generator.py:149-163)."""
        B, T = f0.shape
        cfg = self.cfg
        ws = self._workspace(B, T)
        S = T * cfg.total_upsample
        nh = cfg.nb_harmonics + 1
        if phase_vec is None:
            phase_vec = (torch.rand(B, nh, device=self.device) * 2 - 1) * math.pi
        if noise is None:
            noise = torch.randn(B, nh, S, device=self.device)
        phase_vec = phase_vec.to(self.device, torch.float32).reshape(B, nh).contiguous()
        noise = noise.to(self.device, torch.float32).reshape(B, nh, S).contiguous()
        ops.hift_source(f0.contiguous(), phase_vec, noise, self.lin_w, self.lin_b, ws["src_work"], ws["s"],
                        cfg.total_upsample, float(cfg.sampling_rate), cfg.nsf_alpha, cfg.nsf_sigma, cfg.nsf_voiced_threshold)
        return ws["s"]

    @torch.no_grad()
    def inference(self, speech_feat: torch.Tensor, cache_source: torch.Tensor = torch.zeros(1, 1, 0),
                  phase_vec=None, noise=None):
        """generator.py:399-411 -> (generated_speech (B,S), s (B,1,S))."""
        assert self._loaded
        B, _, T = speech_feat.shape
        ws = self._workspace(B, T)
        ops.to_channels_last(speech_feat.to(self.device, torch.float32).contiguous(), ws["mel_cl"])
        self._f0_from_cl(ws, B, T)
        s = self.source(ws["f0"], phase_vec, noise)
        if cache_source.shape[2] != 0:
            s[:, :cache_source.shape[2]] = cache_source.to(self.device, torch.float32).reshape(B, -1)
        wav = self._decode_cl(ws, B, T, s)
        return wav, s.unsqueeze(1)
