"""CosyVoice-v1 flow ``MaskedDiffWithXvec`` on MI355X (/root/reference/cosyvoice/flow/flow.py:25-160, SURVEY.md §8a row F6;
examples/tts_vc/cosyvoice/conf/cosyvoice.yaml:66-114): speech tokens -> ConformerEncoder -> encoder_proj ->
InterpolateRegulator (prompt / head / middle / tail interpolated separately, flow/length_regulator.py:49-70) = mu ->
ConditionalCFM (flow cache carrying the prompt + last 34 frames of noise and mu between chunks, flow_matching.py:37-70; 10
Euler steps with CFG) over the NON-causal two-level estimator (flow/decoder.py:88-334 with causal=False, channels=[256, 256]:
GroupNorm(8) blocks, stride-2 Downsample1D, ConvTranspose1d(4,2,1) Upsample1D).

Kernels: the conformer layers, transformer blocks, implicit-GEMM convolutions (stride / transposed phases as in HiFT) and the
CFM pack / update launches are the CosyVoice2 flow's; new here are ``cv_groupnorm_cl`` and ``cv_interp_linear_cl``.  The
reference asserts batch 1, so every mask is all ones; the CFG pair runs as R = 2 rows."""
import math
from typing import Dict, Optional

import torch

from . import ops
from .config import FlowV1Config
from .flow import CausalConditionalCFM, ConditionalDecoder, _P, _round_up
from .hift import _ConvT
from .llm_phoneme import _TextEncoder


class ConditionalDecoderV1(ConditionalDecoder):
    """Non-causal two-level U-Net estimator.  Same slot contract as the parent (``__call__``), same ``forward_cl`` driver."""
    fused_all = False   # GroupNorm resnets, strided down / transposed up convs: not what cv_flow_euler_* composes

    def load(self, sd, prefix="decoder.estimator."):
        if self._tcache:   # time embeddings through the time_mlp weights replaced below
            torch.cuda.synchronize()
            self._tcache.clear()
        cfg = self.cfg
        P = _P(sd, self.dtype, self.device)
        names = ([f"{prefix}down_blocks.0", f"{prefix}down_blocks.1"] + [f"{prefix}mid_blocks.{i}" for i in range(cfg.est_mid_blocks)]
                 + [f"{prefix}up_blocks.0", f"{prefix}up_blocks.1"])
        self.blocks = [dict(res=self._load_resnet(P, f"{n}.0", norm_idx=1),
                            tb=[self._load_tblock(P, sd, f"{n}.1.{j}") for j in range(cfg.est_n_blocks)]) for n in names]
        self.down0_w, self.down0_b = P.conv(f"{prefix}down_blocks.0.2.conv.weight"), P.f32(f"{prefix}down_blocks.0.2.conv.bias")
        self.down1_w, self.down1_b = P.conv(f"{prefix}down_blocks.1.2.weight"), P.f32(f"{prefix}down_blocks.1.2.bias")
        self.up0 = _ConvT(sd[f"{prefix}up_blocks.0.2.conv.weight"].detach().float(), sd[f"{prefix}up_blocks.0.2.conv.bias"].detach().float(),
                          2, 1, self.dtype, self.device)
        self.up1_w, self.up1_b = P.conv(f"{prefix}up_blocks.1.2.weight"), P.f32(f"{prefix}up_blocks.1.2.bias")
        self.fin_w, self.fin_b = P.conv(f"{prefix}final_block.block.0.weight"), P.f32(f"{prefix}final_block.block.0.bias")
        self.fin_g, self.fin_be = P.f32(f"{prefix}final_block.block.1.weight"), P.f32(f"{prefix}final_block.block.1.bias")
        self.proj_w, self.proj_b = P.conv(f"{prefix}final_proj.weight"), P.f32(f"{prefix}final_proj.bias")
        self._load_time(P, sd, prefix, names)

    def _level(self, R, T):
        cfg, dt, dev = self.cfg, self.dtype, self.device
        C, inner, ff = cfg.est_channels, cfg.est_inner, cfg.est_channels * cfg.est_ff_mult
        e = lambda *s, dtype=torch.float32: torch.empty(*s, device=dev, dtype=dtype)
        Tp = _round_up(T, 8)
        return dict(T=T, Tp=Tp, c32a=e(R, T, C), c32b=e(R, T, C), h1=e(R, T, C, dtype=dt), x32=e(R, T, C), xn=e(R, T, C, dtype=dt),
                    qk=e(R, T, 2 * inner, dtype=dt), vt=torch.zeros(R, cfg.est_heads, 64, Tp, device=dev, dtype=dt),
                    ao=e(R, T, inner, dtype=dt), ff=e(R, T, ff, dtype=dt), cat=e(R, T, 2 * C, dtype=dt), d=e(R, T, C, dtype=dt),
                    gn=ops.groupnorm_workspace(R, T, cfg.est_groups, dev))

    def _workspace(self, R, T):
        key = (R, T)
        if key not in self._ws:
            cfg = self.cfg
            hi = self._level(R, T)
            hi["lo"] = self._level(R, (T + 1) // 2)      # Conv1d(k3, stride 2, pad 1): ceil(T / 2) frames
            hi["xin"] = torch.empty(R, T, cfg.est_in_channels, device=self.device, dtype=self.dtype)
            hi["v"] = torch.empty(R, T, cfg.output_size, device=self.device)
            self._ws[key] = hi
        return self._ws[key]

    def _resnet(self, rs, ws, R, a_in, lda, cin, tadd):
        """ResnetBlock1D (flow/components/decoder.py:44-59): Conv k3 pad 1 -> GroupNorm -> Mish (+ time term) twice, + 1x1 residual."""
        C, T, G = self.cfg.est_channels, ws["T"], self.cfg.est_groups
        kw = dict(batch=R, a_bs=(T * lda, 0), lda=lda, a_rows=T)
        c32a, c32b = ws["c32a"], ws["c32b"]
        ops.gemm(a_in, rs["w1"], T, C, 3 * cin, cin=cin, tap_base=-1, tap_step=1, bias=rs["b1"], out_f32=c32a,
                 o32_bs=(T * C, 0), ldo32=C, **kw)
        ops.groupnorm_cl(c32a, G, rs["g1"], rs["be1"], 1e-5, ws["gn"], act=ops.ACT_MISH, add=tadd, out_act=ws["h1"])
        ops.gemm(ws["h1"], rs["w2"], T, C, 3 * C, batch=R, a_bs=(T * C, 0), lda=C, a_rows=T, cin=C, tap_base=-1, tap_step=1,
                 bias=rs["b2"], out_f32=c32a, o32_bs=(T * C, 0), ldo32=C)
        ops.groupnorm_cl(c32a, G, rs["g2"], rs["be2"], 1e-5, ws["gn"], act=ops.ACT_MISH, out_f32=c32b)
        ops.gemm(a_in, rs["wr"], T, C, cin, bias=rs["br"], res=c32b, res_bs=(T * C, 0), ldres=C, out_f32=ws["x32"],
                 o32_bs=(T * C, 0), ldo32=C, **kw)

    def _stage(self, blk, ws, R, a_in, lda, cin, tadd, out_act, ldoa):
        self._resnet(blk["res"], ws, R, a_in, lda, cin, tadd)
        n = len(blk["tb"])
        for j, tb in enumerate(blk["tb"]):
            if j < n - 1:
                self._tblock(tb, ws, R, None)
            else:
                self._tblock(tb, ws, R, None, out_act=out_act, ldoa=ldoa)

    def forward_cl(self, ws, R, tadd_row: torch.Tensor, klen=None):
        """ws['xin'] (R,T,320) -> ws['v'] (R,T,80) fp32 (flow/decoder.py:251-334, masks all ones)."""
        assert klen is None, "the v1 estimator runs unpadded (GroupNorm statistics span the whole sequence)"
        cfg = self.cfg
        C, T = cfg.est_channels, ws["T"]
        lo = ws["lo"]
        Td = lo["T"]
        ta = lambda bi: tadd_row[bi * C:(bi + 1) * C]
        nb = len(self.blocks)
        # down 0 @T: skip0 -> cat_hi[:, :, C:]; Downsample1D = Conv1d(k3, stride 2, pad 1) -> lo.d
        self._stage(self.blocks[0], ws, R, ws["xin"], cfg.est_in_channels, cfg.est_in_channels, ta(0), ws["cat"][:, :, C:], 2 * C)
        ops.gemm(ws["cat"][:, :, C:], self.down0_w, Td, C, 3 * C, batch=R, a_bs=(T * 2 * C, 0), lda=2 * C, a_rows=T, cin=C,
                 a_row_stride=2, tap_base=-1, tap_step=1, bias=self.down0_b, out_act=lo["d"], oa_bs=(Td * C, 0), ldoa=C)
        # down 1 @T/2: skip1 -> cat_lo[:, :, C:]; Conv1d(k3, pad 1) -> lo.d
        self._stage(self.blocks[1], lo, R, lo["d"], C, C, ta(1), lo["cat"][:, :, C:], 2 * C)
        ops.gemm(lo["cat"][:, :, C:], self.down1_w, Td, C, 3 * C, batch=R, a_bs=(Td * 2 * C, 0), lda=2 * C, a_rows=Td, cin=C,
                 tap_base=-1, tap_step=1, bias=self.down1_b, out_act=lo["d"], oa_bs=(Td * C, 0), ldoa=C)
        # mid blocks @T/2; the last one lands in cat_lo[:, :, :C] next to skip1
        for bi in range(2, nb - 2):
            last = bi == nb - 3
            self._stage(self.blocks[bi], lo, R, lo["d"], C, C, ta(bi), lo["cat"] if last else lo["d"], 2 * C if last else C)
        # up 0 @T/2 on [x | skip1]; Upsample1D = ConvTranspose1d(4, 2, 1) -> cat_hi[:, :, :C] (frame 2*Td - 1 >= T is dropped)
        self._stage(self.blocks[nb - 2], lo, R, lo["cat"], 2 * C, 2 * C, ta(nb - 2), lo["d"], C)
        up = self.up0
        for r, (wp, ntaps, cr, _) in enumerate(up.phases):
            ops.gemm(lo["d"], wp, Td, C, ntaps * C, batch=R, a_bs=(Td * C, 0), lda=C, a_rows=Td, cin=C, tap_base=cr, tap_step=-1,
                     bias=up.b, out_act=ws["cat"], oa_bs=(T * 2 * C, 0), ldoa=2 * C, out_row_stride=2, out_row_off=r, out_rows=T)
        # up 1 @T on [x | skip0]; Conv1d(k3, pad 1); final Block1D; final_proj
        self._stage(self.blocks[nb - 1], ws, R, ws["cat"], 2 * C, 2 * C, ta(nb - 1), ws["d"], C)
        ops.conv1d_cl(ws["d"], self.up1_w, 3, pad_left=1, bias=self.up1_b, out_act=ws["h1"])
        ops.conv1d_cl(ws["h1"], self.fin_w, 3, pad_left=1, bias=self.fin_b, out_f32=ws["c32a"])
        ops.groupnorm_cl(ws["c32a"], cfg.est_groups, self.fin_g, self.fin_be, 1e-5, ws["gn"], act=ops.ACT_MISH, out_act=ws["h1"])
        ops.linear(ws["h1"].view(R * T, C), self.proj_w, bias=self.proj_b, out_f32=ws["v"].view(R * T, cfg.output_size))
        return ws["v"]


class ConditionalCFM(CausalConditionalCFM):
    """flow_matching.py:22-124: the noise is drawn per call (torch.randn on the host generator, so a seeded run draws the
    reference's CPU values) and the flow cache pins the prompt + overlap part of z and mu between chunks."""

    def __init__(self, cfg, estimator, device):
        self.cfg, self.estimator, self.device = cfg, estimator, device
        self.inference_cfg_rate = cfg.inference_cfg_rate
        self._graphs: Dict[tuple, tuple] = {}
        self.use_graph = False
        self.use_stage_abi = False   # cv_flow_euler_* composes the CosyVoice2 (causal, fused) estimator only


class MaskedDiffWithXvec:
    def __init__(self, cfg: Optional[FlowV1Config] = None, dtype: torch.dtype = torch.float16, device: str = "cuda"):
        self.cfg = cfg or FlowV1Config.full()
        self.dtype, self.device = dtype, torch.device(device)
        self.fp16 = False
        self.input_frame_rate = self.cfg.input_frame_rate
        self.output_size = self.cfg.output_size
        self.encoder = _TextEncoder(self.cfg, dtype, self.device)
        self.encoder.static_chunk_size = 0
        self.decoder = ConditionalCFM(self.cfg, ConditionalDecoderV1(self.cfg, dtype, self.device), self.device)
        self._loaded = False
        self._bufs: Dict[tuple, dict] = {}

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    def half(self):
        return self

    def load_state_dict(self, sd, strict: bool = False):
        ops.drop_graphs(self.decoder._graphs)   # captured Euler loops hold raw pointers of the estimator weights replaced below
        sd = {k: v.detach().to("cpu") for k, v in sd.items()}
        cfg = self.cfg
        P = _P(sd, self.dtype, self.device)
        self.emb_table = P.f32("input_embedding.weight")
        self.spk_w, self.spk_b = P.w("spk_embed_affine_layer.weight"), P.f32("spk_embed_affine_layer.bias")
        self.proj_w, self.proj_b = P.w("encoder_proj.weight"), P.f32("encoder_proj.bias")
        self.encoder.load(sd, prefix="encoder.")
        m = "length_regulator.model"
        self.reg = [dict(w=P.conv(f"{m}.{3 * i}.weight"), b=P.f32(f"{m}.{3 * i}.bias"), g=P.f32(f"{m}.{3 * i + 1}.weight"),
                         be=P.f32(f"{m}.{3 * i + 1}.bias")) for i in range(cfg.reg_layers)]
        n = 3 * cfg.reg_layers
        self.reg_out_w, self.reg_out_b = P.conv(f"{m}.{n}.weight"), P.f32(f"{m}.{n}.bias")
        self.decoder.estimator.load(sd)
        self._loaded = True
        return self

    def mel_len(self, n_tokens: int, sample_rate: int) -> int:
        return int(n_tokens / self.input_frame_rate * sample_rate / self.cfg.hop_size)       # flow.py:143

    def _buffers(self, N, Tm, D):
        key = (N, Tm, D)
        enc, est = self.encoder, self.decoder.estimator      # all per-shape caches of the flow are bounded together
        ops.bound_cache(self._bufs, key, enc._ws, enc._pos, est._ws, self.decoder._graphs, cap=getattr(self, "shape_cache_cap", None))
        if key not in self._bufs:
            dev, O = self.device, self.cfg.output_size
            e = lambda *s, dtype=torch.float32: torch.empty(*s, device=dev, dtype=dtype)
            self._bufs[key] = dict(idx=torch.empty(1, N, device=dev, dtype=torch.int32), emb_in=torch.zeros(1, D, device=dev),
                                   emb_n=torch.zeros(1, _round_up(D, 8), device=dev, dtype=self.dtype), spks=e(1, O),
                                   h=e(N, O), reg_a=e(1, Tm, O, dtype=self.dtype), reg_c=e(1, Tm, O), mu=e(1, Tm, O),
                                   cond=torch.zeros(1, Tm, O, device=dev), x=e(1, Tm, O), mel=e(1, O, Tm),
                                   gn=ops.groupnorm_workspace(1, Tm, self.cfg.reg_groups, dev))
        return self._bufs[key]

    def _regulate(self, bf, n1, n2, mel_len1, mel_len2, sample_rate):
        """InterpolateRegulator.inference (length_regulator.py:49-70): bf['h'] (n1 + n2, 80) fp32 -> bf['mu'] (1, Tm, 80) fp32."""
        cfg = self.cfg
        h, a = bf["h"], bf["reg_a"][0]
        n20 = int(20 / cfg.input_frame_rate * sample_rate / cfg.hop_size)
        segs = []                                        # (token rows, mel rows)
        if n1 != 0:
            segs.append(((0, n1), (0, mel_len1)))
        o = mel_len1
        if n2 > 40:
            segs += [((n1, n1 + 20), (o, o + n20)), ((n1 + 20, n1 + n2 - 20), (o + n20, o + mel_len2 - n20)),
                     ((n1 + n2 - 20, n1 + n2), (o + mel_len2 - n20, o + mel_len2))]
        else:
            segs.append(((n1, n1 + n2), (o, o + mel_len2)))
        for (s0, s1), (d0, d1) in segs:
            ops.interp_linear_cl(h[s0:s1], a[d0:d1])
        for l in self.reg:
            ops.conv1d_cl(bf["reg_a"], l["w"], 3, pad_left=1, bias=l["b"], out_f32=bf["reg_c"])
            ops.groupnorm_cl(bf["reg_c"], cfg.reg_groups, l["g"], l["be"], 1e-5, bf["gn"], act=ops.ACT_MISH, out_act=bf["reg_a"])
        ops.conv1d_cl(bf["reg_a"], self.reg_out_w, 1, bias=self.reg_out_b, out_f32=bf["mu"])

    @torch.no_grad()
    def inference(self, token, token_len, prompt_token, prompt_token_len, prompt_feat, prompt_feat_len, embedding, flow_cache,
                  sample_rate, n_timesteps=10, z=None, return_mu=False):
        """Reference signature (flow.py:108-119) -> (mel (1,80,T2) float32, flow_cache (1,80,T1+34,2) float32).
        ``z`` (1,80,T1+T2) overrides the ``torch.randn_like(mu)`` draw of flow_matching.py:56."""
        assert self._loaded and token.shape[0] == 1
        cfg, dev = self.cfg, self.device
        O = cfg.output_size
        n1, n2 = int(prompt_token.shape[1]), int(token.shape[1])
        N = n1 + n2
        mel_len1, mel_len2 = int(prompt_feat.shape[1]), self.mel_len(n2, sample_rate)
        Tm = mel_len1 + mel_len2
        D = embedding.shape[1]
        bf = self._buffers(N, Tm, D)
        # speaker: F.normalize -> Linear (flow.py:126-127); RMSNorm kernel with scale 1/sqrt(D) == x / ||x||
        bf["emb_in"].copy_(embedding.to(dev, torch.float32))
        ops.layernorm(bf["emb_in"], None, None, 1e-24 / D, rms=True, out_scale=1.0 / math.sqrt(D), out_act=bf["emb_n"][:, :D])
        ops.gemm(bf["emb_n"], self.spk_w, 1, O, D, lda=bf["emb_n"].stride(0), bias=self.spk_b, out_f32=bf["spks"], ldo32=O)
        # tokens -> embedding -> ConformerEncoder -> encoder_proj (flow.py:130-139)
        bf["idx"][:, :n1].copy_(prompt_token.to(dev, torch.int32))
        bf["idx"][:, n1:].copy_(token.to(dev, torch.int32))
        bf["idx"].clamp_(min=0)
        ews = self.encoder._workspace(1, N)
        ops.embedding(self.emb_table, bf["idx"].view(-1), ews["x_in"].view(N, cfg.input_size))
        xa = self.encoder.forward(ews, 1, N)
        ops.linear(xa.view(N, cfg.enc_dim), self.proj_w, bias=self.proj_b, out_f32=bf["h"])
        self._regulate(bf, n1, n2, mel_len1, mel_len2, sample_rate)
        # conditions: prompt mel then zeros (flow.py:145-148)
        bf["cond"].zero_()
        bf["cond"][:, :mel_len1].copy_(prompt_feat.to(dev, torch.float32))
        # ConditionalCFM.forward (flow_matching.py:56-66): noise, cache overwrite of the prompt + overlap part, new cache
        if z is None:
            z = torch.randn(1, O, Tm)
        x, mu = bf["x"], bf["mu"]
        x.copy_(z.to(dev, torch.float32).transpose(1, 2))
        cs = min(int(flow_cache.shape[2]), Tm)
        if cs != 0:
            fc = flow_cache.to(dev, torch.float32)
            x[:, :cs].copy_(fc[:, :, :cs, 0].transpose(1, 2))
            mu[:, :cs].copy_(fc[:, :, :cs, 1].transpose(1, 2))
        keep = lambda v: torch.cat([v[:, :mel_len1], v[:, -34:]], dim=1).transpose(1, 2)
        new_cache = torch.stack([keep(x), keep(mu)], dim=-1).contiguous()
        if return_mu:
            return mu.transpose(1, 2).clone(), new_cache
        self.decoder.solve(x, mu, bf["spks"], bf["cond"], n_timesteps)
        ops.to_channels_first(x, bf["mel"])
        return bf["mel"][:, :, mel_len1:].float().clone(), new_cache
