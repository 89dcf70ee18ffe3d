"""BigVGAN anti-aliased activation on MI355X: drop-in for the reference's fused ``Activation1d``
(/root/reference/cosyvoice/BigVGAN/alias_free_activation/cuda/activation1d.py:36-76), whose CUDA extension
(anti_alias_activation_cuda.cu) is the reference's only native kernel.  Same call contract: ``forward(x [B,C,T])`` with
a Snake / SnakeBeta activation object carrying ``alpha`` (and ``beta``) and ``alpha_logscale``."""
import ctypes as C
import math

import torch

from . import _lib as L


def kaiser_sinc_filter12(cutoff: float = 0.25, half_width: float = 0.3) -> torch.Tensor:
    """12-tap kaiser-windowed sinc of UpSample1d/DownSample1d(ratio 2) — alias_free_activation/torch/filter.py:62-94."""
    k, half = 12, 6
    A = 2.285 * (half - 1) * math.pi * 4 * half_width + 7.95
    beta = 0.1102 * (A - 8.7) if A > 50.0 else (0.5842 * (A - 21) ** 0.4 + 0.07886 * (A - 21.0) if A >= 21.0 else 0.0)
    window = torch.kaiser_window(k, beta=beta, periodic=False)
    time = torch.arange(-half, half) + 0.5
    f = 2 * cutoff * window * torch.sinc(2 * cutoff * time)
    return (f / f.sum()).to(torch.float32)


class Activation1d:
    def __init__(self, activation, up_ratio: int = 2, down_ratio: int = 2, up_kernel_size: int = 12, down_kernel_size: int = 12,
                 device: str = "cuda"):
        if (up_ratio, down_ratio, up_kernel_size, down_kernel_size) != (2, 2, 12, 12):
            raise ValueError("the fused kernel is hard-wired to ratio 2 / 12 taps, as the reference's (activation1d.py:16-19)")
        self.act = activation
        self.device = torch.device(device)
        f = kaiser_sinc_filter12().to(self.device)
        self.up_filter, self.down_filter = f.contiguous(), f.clone().contiguous()

    def _log_params(self):
        alpha = self.act.alpha.detach().to(self.device, torch.float32)
        beta = self.act.beta.detach().to(self.device, torch.float32) if hasattr(self.act, "beta") else alpha  # Snake: beta = alpha
        if not getattr(self.act, "alpha_logscale", False):  # exp is baked into the kernel (activation1d.py:66-71)
            alpha, beta = torch.log(alpha), torch.log(beta)
        return alpha.contiguous(), beta.contiguous()

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("cosyvoice_amd.bigvgan needs device tensors (no CPU path)")
        x = x.contiguous()
        B, Cc, T = x.shape
        y = torch.empty_like(x)
        a, b = self._log_params()
        L.check(L.lib().cv_anti_alias_act(C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), L.TORCH_DT[x.dtype], B, Cc, T,
                                          C.c_void_p(self.up_filter.data_ptr()), C.c_void_p(self.down_filter.data_ptr()),
                                          C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), L.stream_ptr()), "cv_anti_alias_act")
        return y

    __call__ = forward


# ===================================================================================================== full generator
from typing import Dict, Optional  # noqa: E402

from . import ops  # noqa: E402
from .config import BigVGANConfig  # noqa: E402
from .hift import _Conv, _ConvT, _get_padding, _round_up  # noqa: E402
from .weights import fold_weight_norm  # noqa: E402


def anti_alias_act_cl(x, y, C_, up_filter, down_filter, alpha_log, beta_log):
    """x (B,T,ldx) -> y (B,T,ldy), first ``C_`` channels: UpSample1d(2) -> SnakeBeta -> DownSample1d(2), channels-last."""
    B, T, _ = x.shape
    L.check(L.lib().cv_anti_alias_act_cl(C.c_void_p(x.data_ptr()), x.stride(1), L.TORCH_DT[x.dtype], C.c_void_p(y.data_ptr()),
                                         y.stride(1), L.TORCH_DT[y.dtype], B, T, C_, C.c_void_p(up_filter.data_ptr()),
                                         C.c_void_p(down_filter.data_ptr()), C.c_void_p(alpha_log.data_ptr()),
                                         C.c_void_p(beta_log.data_ptr()), L.stream_ptr()), "cv_anti_alias_act_cl")


class BigVGAN:
    """Drop-in for the reference's ``BigVGAN`` generator (/root/reference/cosyvoice/BigVGAN/bigvgan.py:243-438): same
    constructor knobs (as ``BigVGANConfig``), same state-dict keys (legacy weight-norm ``weight_g/weight_v``,
    ``resblocks.{n}.activations.{m}.act.{alpha,beta}`` in log scale), same ``forward(batch, device)`` contract
    -> ``(wav (B,S), (mel_feat_out, None))``.  ``encoder1`` / ``encoder2`` are injected callables ``(x, x_len) -> (y, mask)``
    exactly as in the reference (e.g. ``cosyvoice_amd.flow.UpsampleConformerEncoder``); with both ``None`` the token
    embeddings go straight to ``encoder_proj``.

    Device layout: channels-last (B, T, C) everywhere; the residual stream of every AMP block is fp32, conv operands are
    ``dtype`` (fp32 MFMA by default, as the vocoder is never half-ed by the reference's orchestrator); convs and
    transposed convs are the tap-GEMMs of ``cv_gemm`` (bias, speaker conditioning, block residual, the mean over the three
    parallel AMP blocks and the final tanh all live in GEMM epilogues); the anti-aliased SnakeBeta is ``cv_anti_alias_act_cl``."""

    def __init__(self, cfg: Optional[BigVGANConfig] = None, dtype: torch.dtype = torch.float32, device: str = "cuda",
                 encoder1=None, encoder2=None, f32_products: str = "bf16x3"):
        if not torch.cuda.is_available():
            raise RuntimeError("cosyvoice_amd needs an MI355X (no CPU fallback)")
        self.f32_products = f32_products  # fp32 operands: "exact" f32 MFMA or bf16x3 split products (ops.f32_products)
        self.cfg = cfg or BigVGANConfig.full()
        self.dtype, self.device = dtype, torch.device(device)
        self.encoder1, self.encoder2 = encoder1, encoder2
        self.num_kernels = len(self.cfg.resblock_kernel_sizes)
        self.num_upsamples = len(self.cfg.upsample_rates)
        f = kaiser_sinc_filter12().to(self.device)
        self.up_filter, self.down_filter = f.contiguous(), f.clone().contiguous()
        self._loaded = False
        self._ws: Dict[tuple, dict] = {}

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    def load_state_dict(self, sd, strict: bool = False):
        sd = {k: v.detach().to("cpu", torch.float32) for k, v in sd.items()}
        cfg, dt, dev = self.cfg, self.dtype, self.device
        f32 = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()
        wdt = lambda t: t.to(device=dev, dtype=dt).contiguous()
        self.emb = f32(sd["input_embedding.weight"])
        self.proj_w, self.proj_b = wdt(sd["encoder_proj.weight"]), f32(sd["encoder_proj.bias"])
        self.mel_w, self.mel_b = wdt(sd["mel_proj.weight"]), f32(sd["mel_proj.bias"])
        self.conv_pre = _Conv(fold_weight_norm(sd, "conv_pre"), sd["conv_pre.bias"], dt, dev, pad_left=3)
        self.cond_w = [wdt(sd["cond_layer.weight"][:, :, 0])]
        self.cond_b = [f32(sd["cond_layer.bias"])]
        self.ups, self.blocks = [], []
        for i, (u, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
            self.ups.append(_ConvT(fold_weight_norm(sd, f"ups.{i}.0"), sd[f"ups.{i}.0.bias"], u, (k - u) // 2, dt, dev))
            if cfg.cond_in_each_up_layer:
                self.cond_w.append(wdt(sd[f"conds.{i}.weight"][:, :, 0]))
                self.cond_b.append(f32(sd[f"conds.{i}.bias"]))
            for j, (k2, dils) in enumerate(zip(cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes)):
                name = f"resblocks.{i * self.num_kernels + j}"
                blk = []
                for d_i, d in enumerate(dils):
                    blk.append(dict(
                        c1=_Conv(fold_weight_norm(sd, f"{name}.convs1.{d_i}"), sd[f"{name}.convs1.{d_i}.bias"], dt, dev,
                                 dilation=d, pad_left=_get_padding(k2, d)),
                        c2=_Conv(fold_weight_norm(sd, f"{name}.convs2.{d_i}"), sd[f"{name}.convs2.{d_i}.bias"], dt, dev,
                                 dilation=1, pad_left=_get_padding(k2, 1)),
                        a1=(f32(sd[f"{name}.activations.{2 * d_i}.act.alpha"]), f32(sd[f"{name}.activations.{2 * d_i}.act.beta"])),
                        a2=(f32(sd[f"{name}.activations.{2 * d_i + 1}.act.alpha"]),
                            f32(sd[f"{name}.activations.{2 * d_i + 1}.act.beta"]))))
                self.blocks.append(blk)
        self.act_post = (f32(sd["activation_post.act.alpha"]), f32(sd["activation_post.act.beta"]))
        w_post = fold_weight_norm(sd, "conv_post")                      # (1, ch, 7): N padded to 4 for the vector epilogue
        wp = torch.zeros(4, w_post.shape[1], w_post.shape[2])
        wp[0] = w_post[0]
        bp = torch.zeros(4)
        bp[0] = sd["conv_post.bias"][0]
        self.conv_post = _Conv(wp, bp, dt, dev, pad_left=3)
        self._loaded = True
        return self

    # ------------------------------------------------------------------ workspaces
    def _workspace(self, B, N):
        ws = self._ws.get((B, N))
        if ws is not None:
            return ws
        ops.bound_cache(self._ws, (B, N))
        cfg, dt, dev = self.cfg, self.dtype, self.device
        z = lambda *s, dtype=torch.float32: torch.zeros(*s, device=dev, dtype=dtype)   # zero: padded channels stay finite
        ch = 4 if dt == torch.float32 else 8
        c0 = cfg.upsample_initial_channel
        ws = dict(idx=torch.empty(B * N, device=dev, dtype=torch.int32),
                  emb=z(B, N, _round_up(self.proj_w.shape[1], ch), dtype=dt),   # encoder_proj input (= input_size without encoders)
                  proj=z(B, N, self.conv_pre.cin_pad, dtype=dt),
                  x0=z(B, N, c0), x0a=z(B, N, c0, dtype=dt), mel=z(B, N, cfg.mel_bin),
                  spk=z(_round_up(B, 1), _round_up(cfg.speaker_embedding_dim, ch), dtype=dt),
                  conds=[z(B, w.shape[0]) for w in self.cond_w])
        t, lens, chans = N, [], []
        for i, u in enumerate(cfg.upsample_rates):
            t *= u
            lens.append(t)
            chans.append(c0 // 2 ** (i + 1))
        ws["lens"], ws["chans"] = lens, chans
        for i, (t, c) in enumerate(zip(lens, chans)):
            cp = _round_up(c, ch)
            ws[f"x{i}"] = z(B, t, c)                                   # stage input: ups + speaker conditioning
            ws[f"xa{i}"] = z(B, t, cp, dtype=dt)                        # activated conv operand
            ws[f"t{i}"] = z(B, t, c)                                    # conv1 output (fp32 into the second activation)
            ws[f"ta{i}"] = z(B, t, cp, dtype=dt)
            ws[f"r{i}"] = [z(B, t, c), z(B, t, c)]                      # block residual ping-pong
            ws[f"acc{i}"] = [z(B, t, c), z(B, t, c)]                    # running sum over the parallel AMP blocks
            ws[f"o{i}"] = z(B, t, c)                                    # stage output (mean of the blocks), fp32
            ws[f"oa{i}"] = z(B, t, cp, dtype=dt)                        # ... and as the next transposed conv's operand
        ws["post"] = z(B, lens[-1], 4)
        self._ws[(B, N)] = ws
        return ws

    def _act(self, x, y, c, ab):
        anti_alias_act_cl(x, y, c, self.up_filter, self.down_filter, ab[0], ab[1])

    def _conv(self, c: _Conv, x, **kw):
        ops.conv1d_cl(x, c.w, c.k, dilation=c.dilation, pad_left=c.pad_left, bias=c.b, **kw)

    # ------------------------------------------------------------------ BigVGAN.forward (bigvgan.py:384-438)
    @torch.no_grad()
    def forward(self, batch: dict, device=None):
        with ops.f32_products(self.f32_products):
            return self._forward(batch)

    def _forward(self, batch: dict):
        assert self._loaded
        cfg, dt, dev = self.cfg, self.dtype, self.device
        token = batch["speech_token"].to(dev)
        token_len = batch["speech_token_len"].to(dev)
        emb = batch["embedding"].to(dev, torch.float32)
        B, N0 = token.shape
        # input_embedding(clamp(token, 0)) * mask: padded positions get index -1 = a zero row (cv_embedding)
        valid = torch.arange(N0, device=dev)[None, :] < token_len[:, None]
        idx = torch.where(valid, token.clamp(min=0), torch.full_like(token, -1)).to(torch.int32).reshape(-1).contiguous()
        if self.encoder1 is not None or self.encoder2 is not None:
            x = torch.zeros(B, N0, cfg.input_size, device=dev)
            ops.embedding(self.emb, idx, x.view(B * N0, -1))
            if self.encoder1 is not None:
                x, _ = self.encoder1(x, token_len)
                token_len = token_len * 2
            mel_from_enc = None
            if self.encoder2 is not None:
                x, _ = self.encoder2(x, token_len)
                token_len = token_len * 2
                mel_from_enc = x
            N = x.shape[1]
            ws = self._workspace(B, N)
            ws["emb"][:, :, :x.shape[2]].copy_(x)
        else:
            N = N0
            ws = self._workspace(B, N)
            ops.embedding(self.emb, idx, ws["emb"].view(B * N, -1))
            mel_from_enc = None
        in_dim = self.proj_w.shape[1]
        ops.gemm(ws["emb"], self.proj_w, B * N, cfg.output_size, in_dim, lda=ws["emb"].stride(1), bias=self.proj_b,
                 out_act=ws["proj"].view(B * N, -1), ldoa=ws["proj"].stride(1))
        # speaker conditioning vectors: cond_layer / conds[i] are 1x1 convs of a length-1 signal = linears (:414,:425)
        ws["spk"][:B, :emb.shape[1]].copy_(emb)
        for w, b, out in zip(self.cond_w, self.cond_b, ws["conds"]):
            ops.gemm(ws["spk"], w, B, w.shape[0], w.shape[1], lda=ws["spk"].stride(0), bias=b, out_f32=out, ldo32=out.stride(0))

        def cond_res(k, t):
            c = ws["conds"][k]
            return c.view(B, 1, -1).expand(B, t, c.shape[1])  # stride 0 over time: broadcast residual of the GEMM epilogue

        # conv_pre + cond_layer (:413-414)
        self._conv(self.conv_pre, ws["proj"], res=cond_res(0, N), out_f32=ws["x0"], out_act=ws["x0a"])
        if mel_from_enc is None:
            ops.gemm(ws["x0a"], self.mel_w, B * N, cfg.mel_bin, self.mel_w.shape[1], lda=ws["x0a"].stride(1), bias=self.mel_b,
                     out_f32=ws["mel"].view(B * N, -1), ldo32=cfg.mel_bin)
            mel = ws["mel"]
        else:
            # encoder2-width mel head (:405): the same GEMM, on the encoder output
            enc = mel_from_enc.to(dt).contiguous().view(B * N, -1)
            mel = torch.empty(B, N, cfg.mel_bin, device=dev)
            ops.gemm(enc, self.mel_w, B * N, cfg.mel_bin, self.mel_w.shape[1], lda=enc.stride(0), bias=self.mel_b,
                     out_f32=mel.view(B * N, -1), ldo32=cfg.mel_bin)
        cur_a, t_in = ws["x0a"], N
        nk = self.num_kernels
        for i in range(self.num_upsamples):
            t_out, c = ws["lens"][i], ws["chans"][i]
            up, x32 = self.ups[i], ws[f"x{i}"]
            cres = cond_res(i + 1, t_out) if cfg.cond_in_each_up_layer else None
            for r, (wp, ntaps, cr, _) in enumerate(up.phases):   # ConvTranspose1d as `u` phase GEMMs (+ conds[i], :421-425)
                ops.gemm(cur_a, wp, t_in, c, ntaps * up.cin, batch=B, a_bs=(cur_a.stride(0), 0), lda=cur_a.stride(1),
                         a_rows=t_in, cin=up.cin, tap_base=cr, tap_step=-1, bias=up.b,
                         res=(ws["conds"][i + 1] if cres is not None else None),
                         res_bs=((ws["conds"][i + 1].stride(0), 0) if cres is not None else (0, 0)), ldres=0,
                         out_f32=x32, o32_bs=(x32.stride(0), 0), ldo32=c, out_row_stride=up.u, out_row_off=r, out_rows=t_out)
            # three parallel AMPBlock1, mean (:427-434); AMPBlock1.forward :128-137
            acc, xa, t32, ta, rr = ws[f"acc{i}"], ws[f"xa{i}"], ws[f"t{i}"], ws[f"ta{i}"], ws[f"r{i}"]
            for j in range(nk):
                blk = self.blocks[i * nk + j]
                cur32 = x32
                for d_i, lay in enumerate(blk):
                    self._act(cur32, xa, c, lay["a1"])
                    self._conv(lay["c1"], xa, out_f32=t32)
                    self._act(t32, ta, c, lay["a2"])
                    if d_i < len(blk) - 1:
                        nxt = rr[d_i & 1]
                        self._conv(lay["c2"], ta, res=cur32, out_f32=nxt)
                        cur32 = nxt
                    else:
                        fin = dict(res=cur32)
                        if j > 0:
                            fin["res2"] = acc[(j - 1) & 1]
                        if j < nk - 1:
                            fin["out_f32"] = acc[j & 1]
                        else:
                            fin.update(out_scale=1.0 / nk, out_f32=ws[f"o{i}"], out_act=ws[f"oa{i}"])
                        self._conv(lay["c2"], ta, **fin)
            cur_a, t_in = ws[f"oa{i}"], t_out
        # activation_post -> conv_post -> tanh (:436-441)
        last = self.num_upsamples - 1
        self._act(ws[f"o{last}"], ws[f"xa{last}"], ws["chans"][last], self.act_post)
        if dt == torch.float32:
            self._conv(self.conv_post, ws[f"xa{last}"], act=ops.ACT_TANH, out_act=ws["post"])  # tanh in the GEMM epilogue
            wav = ws["post"][:, :, 0].clone()
        else:
            # the epilogue's activated output has the operand type; a 16-bit waveform would throw away the fp32 accumulate
            self._conv(self.conv_post, ws[f"xa{last}"], out_f32=ws["post"])
            wav = torch.tanh(ws["post"][:, :, 0])
        return wav, (mel.clone(), None)

    __call__ = forward

    def inference(self, speech_token, speech_token_len, embedding):
        wav, _ = self.forward(dict(speech_token=speech_token, speech_token_len=speech_token_len, embedding=embedding))
        return wav
