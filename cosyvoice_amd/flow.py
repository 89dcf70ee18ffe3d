"""Flow-matching mel decoder on MI355X — host side of the drop-in for the reference's
``CausalMaskedDiffWithXvec`` (/root/reference/cosyvoice/flow/flow.py:163-319), its
``UpsampleConformerEncoder`` (transformer/upsample_encoder.py:100-318), ``CausalConditionalCFM``
(flow/flow_matching.py:209-240) and the estimator ``ConditionalDecoder`` (flow/decoder.py:88-334).

Same attribute surface the orchestrator touches: ``.fp16``, ``.input_frame_rate``,
``.token_mel_ratio``, ``.pre_lookahead_len``, ``.encoder.static_chunk_size``,
``.decoder.estimator(.static_chunk_size)``, ``inference(...) -> (mel (1,80,T_g) fp32, None)``; same state-dict
key names.  All arithmetic runs in libcosyvoice_amd.so: channels-last activations, fp32 residual streams,
16-bit (bf16 / fp16) MFMA operands with fp32 accumulation, flash attention, fused epilogues.

Beyond the reference (which is batch-1 only, flow.py:277): ``inference_batch`` runs B equal-shape utterances in
one pass (the CFG pair of every utterance is batched: 2B sequences per estimator call).
"""
import math
import os
from typing import Dict, List, Optional

import torch

from . import _lib as L
from . import ops
from .config import FlowConfig


def _round_up(x, m):
    return (x + m - 1) // m * m


class _P:
    """Device-resident packed parameters, fetched from a state dict by reference key."""

    def __init__(self, sd, dtype, device):
        self.sd, self.dt, self.dev = sd, dtype, device

    def f32(self, key):
        return self.sd[key].detach().to(device=self.dev, dtype=torch.float32).contiguous()

    def w(self, key):
        return self.sd[key].detach().to(torch.float32).to(device=self.dev, dtype=self.dt).contiguous()

    def conv(self, key):
        """Conv1d weight (Cout,Cin,k) -> (Cout, k*Cin), k = tap*Cin + ci."""
        w = self.sd[key].detach().to(torch.float32)
        return w.permute(0, 2, 1).reshape(w.shape[0], -1).to(device=self.dev, dtype=self.dt).contiguous()


# =============================================================================== encoder
class UpsampleConformerEncoder:
    def __init__(self, cfg: FlowConfig, dtype, device):
        self.cfg, self.dtype, self.device = cfg, dtype, device
        self.static_chunk_size = 0
        self._ws: Dict[tuple, dict] = {}
        self._pos: Dict[tuple, torch.Tensor] = {}

    def output_size(self):
        return self.cfg.enc_dim

    def _load_layer(self, P, sd, name, norm_mha="norm_mha", norm_ff="norm_ff"):
        """One ConformerEncoderLayer (rel-pos self-attention + FFN, no cnn module, no macaron): packed operands.  The
        TransformerEncoderLayer of the v1 LM has the same tensors under the names norm1 / norm2."""
        a = f"{name}.self_attn."
        bq = sd[a + "linear_q.bias"].float()
        wq = sd[a + "linear_q.weight"].float()
        wqqk = torch.cat([wq, wq, sd[a + "linear_k.weight"].float()], 0)
        bqqk = torch.cat([bq + sd[a + "pos_bias_u"].float().reshape(-1), bq + sd[a + "pos_bias_v"].float().reshape(-1),
                          sd[a + "linear_k.bias"].float()], 0)
        # softmax rows sum to 1, so P.(V + 1 b_v^T) = P.V + b_v: the value bias is folded into linear_out's bias
        wout = sd[a + "linear_out.weight"].float()
        bout = sd[a + "linear_out.bias"].float() + wout @ sd[a + "linear_v.bias"].float()
        return dict(wqqk=wqqk.to(device=self.device, dtype=self.dtype).contiguous(),
                    bqqk=bqqk.to(device=self.device).contiguous(), wv=P.w(a + "linear_v.weight"),
                    wpos=P.w(a + "linear_pos.weight"), wout=P.w(a + "linear_out.weight"),
                    bout=bout.to(device=self.device).contiguous(),
                    w1=P.w(f"{name}.feed_forward.w_1.weight"), b1=P.f32(f"{name}.feed_forward.w_1.bias"),
                    w2=P.w(f"{name}.feed_forward.w_2.weight"), b2=P.f32(f"{name}.feed_forward.w_2.bias"),
                    g_mha=P.f32(f"{name}.{norm_mha}.weight"), b_mha=P.f32(f"{name}.{norm_mha}.bias"),
                    g_ff=P.f32(f"{name}.{norm_ff}.weight"), b_ff=P.f32(f"{name}.{norm_ff}.bias"))

    @staticmethod
    def _load_embed(P, name):
        return dict(w=P.w(f"{name}.out.0.weight"), b=P.f32(f"{name}.out.0.bias"), g=P.f32(f"{name}.out.1.weight"),
                    beta=P.f32(f"{name}.out.1.bias"))

    def _invalidate(self):
        """Called by every load(): the projected position tables are products of the layers' linear_pos weights."""
        if self._pos:
            torch.cuda.synchronize()
            self._pos.clear()

    def load(self, sd, prefix="encoder."):
        self._invalidate()
        cfg = self.cfg
        P = _P(sd, self.dtype, self.device)
        D = cfg.enc_dim

        layer = lambda name: self._load_layer(P, sd, name)
        emb = lambda name: self._load_embed(P, name)

        self.embed = emb(prefix + "embed")
        self.up_embed = emb(prefix + "up_embed")
        self.pl1_w, self.pl1_b = P.conv(prefix + "pre_lookahead_layer.conv1.weight"), P.f32(prefix + "pre_lookahead_layer.conv1.bias")
        self.pl2_w, self.pl2_b = P.conv(prefix + "pre_lookahead_layer.conv2.weight"), P.f32(prefix + "pre_lookahead_layer.conv2.bias")
        self.layers = [layer(f"{prefix}encoders.{i}") for i in range(cfg.enc_blocks)]
        self.up_layers = [layer(f"{prefix}up_encoders.{i}") for i in range(cfg.enc_up_blocks)]
        # Upsample1D (nearest x2, left pad 4, conv k5; upsample_encoder.py:59-63) as two 3-tap phase convs:
        #   out[2q]   = (w0+w1) x[q-2] + (w2+w3) x[q-1] + w4 x[q];  out[2q+1] = w0 x[q-2] + (w1+w2) x[q-1] + (w3+w4) x[q]
        w = sd[prefix + "up_layer.conv.weight"].float()  # (D, D, 5)
        ph0 = torch.stack([w[:, :, 0] + w[:, :, 1], w[:, :, 2] + w[:, :, 3], w[:, :, 4]], dim=1)
        ph1 = torch.stack([w[:, :, 0], w[:, :, 1] + w[:, :, 2], w[:, :, 3] + w[:, :, 4]], dim=1)
        self.up_w = [p.reshape(D, 3 * D).to(device=self.device, dtype=self.dtype).contiguous() for p in (ph0, ph1)]
        self.up_b = P.f32(prefix + "up_layer.conv.bias")
        self.after_g, self.after_b = P.f32(prefix + "after_norm.weight"), P.f32(prefix + "after_norm.bias")

    # ---- rel-pos table (EspnetRelPositionalEncoding, embedding.py:220-294) projected by each layer's linear_pos
    def _pos_proj(self, layers, tag, T):
        key = (tag, T)
        if key in self._pos:
            return self._pos[key]
        D = self.cfg.enc_dim
        pos = torch.arange(T - 1, -T, -1, dtype=torch.float32).unsqueeze(1)
        div = torch.exp(torch.arange(0, D, 2, dtype=torch.float32) * -(math.log(10000.0) / D))
        pe = torch.zeros(2 * T - 1, D)
        pe[:, 0::2] = torch.sin(pos * div)
        pe[:, 1::2] = torch.cos(pos * div)
        pe = pe.to(device=self.device, dtype=self.dtype).contiguous()
        outs = []
        for l in layers:
            p = torch.empty(2 * T - 1, D, device=self.device, dtype=self.dtype)
            ops.linear(pe, l["wpos"], out_act=p)
            outs.append(p)
        self._pos[key] = outs
        return outs

    def _workspace(self, R, N):
        key = (R, N)
        if key in self._ws:
            return self._ws[key]
        cfg, dt, dev = self.cfg, self.dtype, self.device
        D, H, U = cfg.enc_dim, cfg.enc_heads, cfg.enc_linear_units
        e = lambda *s, dtype=torch.float32: torch.empty(*s, device=dev, dtype=dtype)
        ws = {}
        for tag, T in (("a", N), ("b", 2 * N)):
            Tp = _round_up(T, 8)
            ldb = _round_up(2 * T - 1, 4)
            ws[tag] = dict(T=T, Tp=Tp, ldb=ldb, xs=e(R, T, D), xn=e(R, T, D, dtype=dt), lin=e(R, T, D),
                           q=e(R, T, 3 * D, dtype=dt), vt=torch.zeros(R, H, 64, Tp, device=dev, dtype=dt),
                           bd=e(R, H, T, ldb), ao=e(R, T, D, dtype=dt), ff=e(R, T, U, dtype=dt), xa=e(R, T, D, dtype=dt))
        ws["tok"] = e(R, N, D, dtype=dt)
        ws["t1"] = e(R, N, D, dtype=dt)
        ws["up"] = e(R, 2 * N, D, dtype=dt)
        ws["mu"] = e(R, 2 * N, cfg.output_size)
        self._ws[key] = ws
        return ws

    def _layer(self, l, w, R, p_l, chunk, klen, last_act=None, act=None, causal=False):
        cfg = self.cfg
        D, H, U = cfg.enc_dim, cfg.enc_heads, cfg.enc_linear_units
        T, Tp, ldb = w["T"], w["Tp"], w["ldb"]
        xs = w["xs"]
        xs2 = xs.view(R * T, D)
        ops.layernorm(xs2, l["g_mha"], l["b_mha"], 1e-12, out_act=w["xn"].view(R * T, D))
        # [q+u | q+v | k] row-major in one GEMM (ld 3D); V^T as the swapped product Wv . xn^T
        ops.linear(w["xn"].view(R * T, D), l["wqqk"], bias=l["bqqk"], out_act=w["q"].view(R * T, 3 * D))
        ops.gemm(l["wv"], w["xn"], D, T, D, batch=R, lda=D, w_bs=(T * D, 0), ldw=D, out_act=w["vt"], oa_bs=(D * Tp, 0), ldoa=Tp)
        # matrix_bd = (q + pos_bias_v) . p^T per head, pre-scaled by 1/sqrt(dk) (attention.py:317-327)
        scale = 1.0 / math.sqrt(D // H)
        ops.gemm(w["q"][:, :, D:], p_l, T, 2 * T - 1, 64, batch=R * H, batch_inner=H, a_bs=(64, T * 3 * D), lda=3 * D,
                 w_bs=(64, 0), ldw=D, out_scale=scale, out_f32=w["bd"], o32_bs=(T * ldb, H * T * ldb), ldo32=ldb)
        ops.attention(w["q"], w["q"][:, :, 2 * D:], w["vt"], w["ao"], B=R, H=H, Hkv=H, Tq=T, Tk=T, scale=scale, q_bs=T * 3 * D,
                      ldq=3 * D, k_bs=T * 3 * D, ldk=3 * D, vt_ld=Tp, o_bs=T * D, ldo=D, klen=klen, chunk=chunk, causal=causal,
                      bias=w["bd"].view(-1)[T - 1:], bias_bs=H * T * ldb, bias_hs=T * ldb, bias_ld=ldb - 1)
        ops.linear(w["ao"].view(R * T, D), l["wout"], bias=l["bout"], res=xs2, out_f32=xs2)
        ops.layernorm(xs2, l["g_ff"], l["b_ff"], 1e-12, out_act=w["xn"].view(R * T, D))
        ffn_act = dict(act=ops.ACT_SILU) if act is None else act      # conformer FFN: swish; v1 TransformerEncoderLayer: ReLU
        ops.linear(w["xn"].view(R * T, D), l["w1"], bias=l["b1"], out_act=w["ff"].view(R * T, U), **ffn_act)
        ops.linear(w["ff"].view(R * T, U), l["w2"], bias=l["b2"], res=xs2, out_f32=xs2,
                   out_act=(last_act.view(R * T, D) if last_act is not None else None))

    def _embed(self, e, x_act2d, w, R):
        T, D = w["T"], self.cfg.enc_dim
        ops.linear(x_act2d, e["w"], bias=e["b"], out_f32=w["lin"].view(R * T, D))
        ops.layernorm(w["lin"].view(R * T, D), e["g"], e["beta"], 1e-5, out_scale=math.sqrt(D), out_f32=w["xs"].view(R * T, D),
                      out_act=w["xa"].view(R * T, D))

    def forward_tokens(self, tok_emb_act, R, N, klen=None, n_valid=None):
        """tok_emb_act: (R,N,512) token embeddings in the operand dtype (already in ws['tok']).
        Returns ws (ws['b']['xa'] holds after_norm output, 16-bit, (R,2N,512)).  upsample_encoder.py:237-304."""
        cfg = self.cfg
        D = cfg.enc_dim
        ws = self._workspace(R, N)
        wa, wb = ws["a"], ws["b"]
        self._embed(self.embed, tok_emb_act.view(R * N, D), wa, R)
        if n_valid is not None:
            # padded call (length bucket / ragged batch): positions beyond a sequence's real end must read as the zero padding the
            # look-ahead conv of the reference sees there (upsample_encoder.py:81-88); everything after it only looks left or is
            # masked by klen
            for r, nv in enumerate(n_valid if isinstance(n_valid, (list, tuple)) else [n_valid] * R):
                if nv < N:
                    wa["xs"][r, nv:].zero_()
                    wa["xa"][r, nv:].zero_()
        # PreLookaheadLayer (upsample_encoder.py:81-96): conv k4 looking right (zero beyond the end), leaky 0.01,
        # conv k3 looking left, residual
        ops.conv1d_cl(wa["xa"], self.pl1_w, cfg.pre_lookahead_len + 1, pad_left=0, bias=self.pl1_b, act=ops.ACT_LEAKY,
                      act_slope=0.01, out_act=ws["t1"])
        ops.conv1d_cl(ws["t1"], self.pl2_w, 3, pad_left=2, bias=self.pl2_b, res=wa["xs"], out_f32=wa["xs"])
        pa = self._pos_proj(self.layers, "a", N)
        for i, l in enumerate(self.layers):
            self._layer(l, wa, R, pa[i], self.static_chunk_size, klen, last_act=(wa["xa"] if i == len(self.layers) - 1 else None))
        # Upsample1D as two phase convs writing interleaved rows
        for r in range(2):
            ops.gemm(wa["xa"], self.up_w[r], N, D, 3 * D, batch=R, a_bs=(N * D, 0), lda=D, a_rows=N, cin=D, tap_base=-2,
                     tap_step=1, bias=self.up_b, out_act=ws["up"], oa_bs=(2 * N * D, 0), ldoa=D, out_row_stride=2,
                     out_row_off=r, out_rows=2 * N)
        self._embed(self.up_embed, ws["up"].view(R * 2 * N, D), wb, R)
        pb = self._pos_proj(self.up_layers, "b", 2 * N)
        klen2 = (klen * 2) if klen is not None else None
        for i, l in enumerate(self.up_layers):
            self._layer(l, wb, R, pb[i], self.static_chunk_size * 2, klen2)
        T2 = 2 * N
        ops.layernorm(wb["xs"].view(R * T2, D), self.after_g, self.after_b, 1e-5, out_act=wb["xa"].view(R * T2, D))
        return ws



    # ---- streaming: chunk-causal encoder cache (upsample_encoder.py:273-296 + utils/mask.py:127-200) -------------
    # Under CosyVoice2Model's static chunk mask (50 tokens / 100 frames, cli/model.py:312-315) a position only attends to its own and
    # earlier chunks, the look-ahead conv reaches 3 tokens to the right and every other op is pointwise or looks left.  So once
    # n tokens are known, everything the encoder computes for positions < s = floor((n - 3) / chunk) * chunk — every layer's keys
    # and values included — is final: a streaming request re-runs the encoder only on rows [s, n) of each chunk call, against
    # per-layer K / V^T kept from the previous calls.  The reference re-encodes the whole sequence per chunk (cli/model.py:380-407);
    # the result is the same up to the fp32 order of a score's scale-and-shift (a tile may be an "interior" tile in one call and a
    # "masked" one in the other).
    def new_stream_cache(self, cap_tokens: int = 1024):
        return EncoderStreamCache(self, cap_tokens)

    def _layer_cached(self, l, lc, loc, p_l, s, n, chunk, klen, last_act=None):
        """One conformer layer on rows [s, n) (local tensors `loc`, M = n - s rows) against the layer cache `lc` = (q rows
        [q+u | q+v | k] (cap, 3D), V^T (H, 64, cap_p)) that already holds rows [0, s) and receives rows [s, n)."""
        cfg = self.cfg
        D, H, U = cfg.enc_dim, cfg.enc_heads, cfg.enc_linear_units
        M = n - s
        qb, vtb = lc
        capp = vtb.shape[2]
        xs = loc["xs"][:M]
        ops.layernorm(xs, l["g_mha"], l["b_mha"], 1e-12, out_act=loc["xn"][:M])
        ops.linear(loc["xn"][:M], l["wqqk"], bias=l["bqqk"], out_act=qb[s:n])
        # V^T of the new rows: through an 8-column-aligned scratch (the cache columns start at s = 50 k: not store-aligned)
        Mp = _round_up(M, 8)
        vt_new = loc["vt"][:D * Mp].view(D, Mp)
        ops.gemm(l["wv"], loc["xn"], D, M, D, batch=1, lda=D, ldw=D, out_act=vt_new, ldoa=Mp)
        vtb.view(D, capp)[:, s:n].copy_(vt_new[:, :M])
        scale = 1.0 / math.sqrt(D // H)
        ldb = _round_up(2 * n - 1, 4)
        bd = loc["bd"].view(-1)[:H * M * ldb].view(H, M, ldb)
        ops.gemm(qb[s:, D:], p_l, M, 2 * n - 1, 64, batch=H, batch_inner=H, a_bs=(64, 0), lda=3 * D, w_bs=(64, 0), ldw=D,
                 out_scale=scale, out_f32=bd, o32_bs=(M * ldb, 0), ldo32=ldb)
        ops.attention(qb[s:], qb[:, 2 * D:], vtb, loc["ao"], B=1, H=H, Hkv=H, Tq=M, Tk=n, scale=scale, q_bs=0, ldq=3 * D, k_bs=0,
                      ldk=3 * D, vt_ld=capp, o_bs=0, ldo=D, klen=klen, chunk=chunk, q_off=s,
                      bias=bd.view(-1)[n - 1 - s:], bias_bs=0, bias_hs=M * ldb, bias_ld=ldb - 1)
        ops.linear(loc["ao"][:M], l["wout"], bias=l["bout"], res=xs, out_f32=xs)
        ops.layernorm(xs, l["g_ff"], l["b_ff"], 1e-12, out_act=loc["xn"][:M])
        ops.linear(loc["xn"][:M], l["w1"], bias=l["b1"], out_act=loc["ff"][:M], act=ops.ACT_SILU)
        ops.linear(loc["ff"][:M], l["w2"], bias=l["b2"], res=xs, out_f32=xs, out_act=last_act)

    def forward_tokens_cached(self, ec, tok_emb_act, N, Nv, klen=None):
        """tok_emb_act (1, N, D): embeddings of all N (>= Nv real, rest padding) tokens.  Encodes rows [ec.n_done, N) and returns the
        after_norm output of ALL 2N frames as ec.xa2[:2N] (16-bit); afterwards ec.n_done = the new stable prefix."""
        cfg = self.cfg
        D, U, H = cfg.enc_dim, cfg.enc_linear_units, cfg.enc_heads
        chunk = int(self.static_chunk_size)
        assert chunk > 0 and N <= ec.cap
        dt, dev = self.dtype, self.device
        s = min(ec.n_done, (max(Nv - cfg.pre_lookahead_len, 0) // chunk) * chunk)
        M, c0 = N - s, max(s - 2, 0)          # rows to encode; first row the look-ahead convs need (2 rows of left context)
        ML = N - c0
        e = lambda *sh, dtype=torch.float32: torch.empty(*sh, device=dev, dtype=dtype)
        # ---- N domain: embed rows [c0, N) -> look-ahead convs -> rows [s, N)
        lin, xs0, xa0 = e(ML, D), e(ML, D), e(ML, D, dtype=dt)
        ops.linear(tok_emb_act.view(N, D)[c0:], self.embed["w"], bias=self.embed["b"], out_f32=lin)
        ops.layernorm(lin, self.embed["g"], self.embed["beta"], 1e-5, out_scale=math.sqrt(D), out_f32=xs0, out_act=xa0)
        if Nv < N:   # padding rows read as the zero padding the look-ahead conv of the reference sees beyond the end
            xs0[Nv - c0:].zero_()
            xa0[Nv - c0:].zero_()
        t1 = e(ML, D, dtype=dt)
        ops.conv1d_cl(xa0.view(1, ML, D), self.pl1_w, cfg.pre_lookahead_len + 1, pad_left=0, bias=self.pl1_b, act=ops.ACT_LEAKY,
                      act_slope=0.01, out_act=t1.view(1, ML, D))
        loc = dict(xs=e(M, D), xn=e(M, D, dtype=dt), ao=e(M, D, dtype=dt), ff=e(M, U, dtype=dt), vt=e(D * _round_up(M, 8), dtype=dt),
                   bd=e(H * M * _round_up(2 * N - 1, 4)))
        if c0 == s - 2:   # conv2 output row s + q reads t1 rows s - 2 + q ..+2: a k3 conv without padding over the slice
            ops.conv1d_cl(t1.view(1, ML, D), self.pl2_w, 3, pad_left=0, T_out=M, bias=self.pl2_b, res=xs0[2:].view(1, M, D),
                          out_f32=loc["xs"].view(1, M, D))
        else:             # s < 2 (s == 0): the sequence start, left zero padding as in the full pass
            ops.conv1d_cl(t1.view(1, ML, D), self.pl2_w, 3, pad_left=2, bias=self.pl2_b, res=xs0.view(1, ML, D),
                          out_f32=loc["xs"].view(1, M, D))
        pa = self._pos_proj(self.layers, "a", N)
        for i, l in enumerate(self.layers):
            self._layer_cached(l, ec.a[i], loc, pa[i], s, N, chunk, klen,
                               last_act=(ec.xa_last[2 + s:2 + N] if i == len(self.layers) - 1 else None))
        # ---- upsample (two 3-tap phase convs over xa rows q - 2 .. q; ec.xa_last has two zero rows in front) -> frames [2s, 2N)
        M2, s2, N2 = 2 * M, 2 * s, 2 * N
        up = e(M2, D, dtype=dt)
        for r in range(2):
            ops.gemm(ec.xa_last[s:], self.up_w[r], M, D, 3 * D, batch=1, lda=D, a_rows=M + 2, cin=D, tap_base=0, tap_step=1,
                     bias=self.up_b, out_act=up, ldoa=D, out_row_stride=2, out_row_off=r, out_rows=M2)
        lin2 = e(M2, D)
        loc2 = dict(xs=e(M2, D), xn=e(M2, D, dtype=dt), ao=e(M2, D, dtype=dt), ff=e(M2, U, dtype=dt), vt=e(D * _round_up(M2, 8), dtype=dt),
                    bd=e(H * M2 * _round_up(2 * N2 - 1, 4)))
        ops.linear(up, self.up_embed["w"], bias=self.up_embed["b"], out_f32=lin2)
        ops.layernorm(lin2, self.up_embed["g"], self.up_embed["beta"], 1e-5, out_scale=math.sqrt(D), out_f32=loc2["xs"])
        pb = self._pos_proj(self.up_layers, "b", N2)
        klen2 = (klen * 2) if klen is not None else None
        for i, l in enumerate(self.up_layers):
            self._layer_cached(l, ec.b[i], loc2, pb[i], s2, N2, 2 * chunk, klen2)
        ops.layernorm(loc2["xs"], self.after_g, self.after_b, 1e-5, out_act=ec.xa2[s2:N2])
        ec.n_done = (max(Nv - cfg.pre_lookahead_len, 0) // chunk) * chunk
        ec.rows_encoded += M
        ec.calls += 1
        return ec.xa2[:N2]


class EncoderStreamCache:
    """Per-request state of UpsampleConformerEncoder.forward_tokens_cached: for each of the 6 + 4 layers the [q+u | q+v | k] rows and V^T
    of every position seen so far, the N-domain output rows the upsampling conv looks back on, and the after_norm output of every frame."""

    def __init__(self, enc: "UpsampleConformerEncoder", cap_tokens: int):
        cfg, dt, dev = enc.cfg, enc.dtype, enc.device
        D, H = cfg.enc_dim, cfg.enc_heads
        self.cap = int(cap_tokens)
        z = lambda *sh: torch.zeros(*sh, device=dev, dtype=dt)
        capp, capp2 = _round_up(self.cap, 8), _round_up(2 * self.cap, 8)
        self.a = [(z(self.cap, 3 * D), z(H, 64, capp)) for _ in enc.layers]
        self.b = [(z(2 * self.cap, 3 * D), z(H, 64, capp2)) for _ in enc.up_layers]
        self.xa_last = z(self.cap + 2, D)          # logical row r at index r + 2 (two rows of left zero padding)
        self.xa2 = z(2 * self.cap, D)              # after_norm output, all frames
        self.n_done = 0                            # tokens whose encoding is final
        self.rows_encoded, self.calls = 0, 0       # diagnostics: token rows run through the layers / calls


# =============================================================================== estimator
class ConditionalDecoder:
    """The CFM estimator (flow/decoder.py:88-334, channels=[256], causal).  ``__call__`` keeps the reference slot
    contract (x,mask,mu,t,spks,cond -> (2,80,T), SURVEY.md §8b level 2); the solver drives ``forward_cl``."""

    def __init__(self, cfg: FlowConfig, dtype, device):
        self.cfg, self.dtype, self.device = cfg, dtype, device
        self.static_chunk_size = 0
        self._ws: Dict[tuple, dict] = {}
        self._tcache: Dict[tuple, torch.Tensor] = {}

    def _load_resnet(self, P, name, norm_idx=2):
        n = norm_idx   # Block1D: Conv1d, GroupNorm, Mish (norm at .1); CausalBlock1D has a Transpose in between (norm at .2)
        return dict(w1=P.conv(f"{name}.block1.block.0.weight"), b1=P.f32(f"{name}.block1.block.0.bias"),
                    g1=P.f32(f"{name}.block1.block.{n}.weight"), be1=P.f32(f"{name}.block1.block.{n}.bias"),
                    w2=P.conv(f"{name}.block2.block.0.weight"), b2=P.f32(f"{name}.block2.block.0.bias"),
                    g2=P.f32(f"{name}.block2.block.{n}.weight"), be2=P.f32(f"{name}.block2.block.{n}.bias"),
                    wr=P.conv(f"{name}.res_conv.weight"), br=P.f32(f"{name}.res_conv.bias"))

    @property
    def fused(self):
        """The row-block kernels (cv_tblock_head / cv_tblock_tail) cover the widths of the reference wiring (256 channels, 8 x 64
        heads, 4 x FFN: cosyvoice2 yaml :68-78) in 16-bit operands; anything else keeps the cv_gemm / cv_layernorm launches.
        CV_FLOW_FUSED=0 forces the unfused launches (A/B measurements, cross-check tests)."""
        cfg = self.cfg
        return (os.environ.get("CV_FLOW_FUSED", "1") != "0" and self.dtype in (torch.float16, torch.bfloat16)
                and cfg.est_channels == 256 and cfg.est_inner == 512 and cfg.est_head_dim == 64 and cfg.est_ff_mult == 4)

    @property
    def fused_all(self):
        """Every block runs on the row-block kernels (what cv_flow_euler_* composes): fused widths and packed resnet weights."""
        return (self.fused and os.environ.get("CV_FLOW_FUSED_RESNET", "1") != "0"
                and all("w1_p" in blk["res"] and blk["res"]["wr"].shape[1] in (256, 320, 512) for blk in self.blocks))

    def _load_tblock(self, P, sd, name):
        wqk = torch.cat([sd[f"{name}.attn1.to_q.weight"].float(), sd[f"{name}.attn1.to_k.weight"].float()], 0)
        tb = dict(g1=P.f32(f"{name}.norm1.weight"), b1=P.f32(f"{name}.norm1.bias"),
                  wqk=wqk.to(device=self.device, dtype=self.dtype).contiguous(), wv=P.w(f"{name}.attn1.to_v.weight"),
                  wo=P.w(f"{name}.attn1.to_out.0.weight"), bo=P.f32(f"{name}.attn1.to_out.0.bias"),
                  g3=P.f32(f"{name}.norm3.weight"), b3=P.f32(f"{name}.norm3.bias"),
                  wf1=P.w(f"{name}.ff.net.0.proj.weight"), bf1=P.f32(f"{name}.ff.net.0.proj.bias"),
                  wf2=P.w(f"{name}.ff.net.2.weight"), bf2=P.f32(f"{name}.ff.net.2.bias"))
        if self.fused:
            # fragment-ordered copies for the row-block kernels: every wave-load of a weight fragment is 1 KiB contiguous
            tb.update(wqkv_p=ops.pack_skinny(torch.cat([tb["wqk"], tb["wv"]], 0).contiguous()), wo_p=ops.pack_skinny(tb["wo"]),
                      wf1_p=ops.pack_skinny(tb["wf1"]), wf2_p=ops.pack_skinny(tb["wf2"]))
        return tb

    def _load_time(self, P, sd, prefix, names):
        """time-embedding path (input independent): TimestepEmbedding + every resnet's Mish->Linear, stacked"""
        self.t1_w, self.t1_b = P.w(f"{prefix}time_mlp.linear_1.weight"), P.f32(f"{prefix}time_mlp.linear_1.bias")
        self.t2_w, self.t2_b = P.w(f"{prefix}time_mlp.linear_2.weight"), P.f32(f"{prefix}time_mlp.linear_2.bias")
        self.tm_w = torch.cat([sd[f"{n}.0.mlp.1.weight"].float() for n in names], 0).to(device=self.device, dtype=self.dtype).contiguous()
        self.tm_b = torch.cat([sd[f"{n}.0.mlp.1.bias"].float() for n in names], 0).to(device=self.device).contiguous()

    def load(self, sd, prefix="decoder.estimator."):
        if self._tcache:   # time embeddings through the time_mlp weights replaced below
            torch.cuda.synchronize()
            self._tcache.clear()
        cfg = self.cfg
        P = _P(sd, self.dtype, self.device)
        self.P = P
        resnet = lambda name: self._load_resnet(P, name)
        tblock = lambda name: self._load_tblock(P, sd, name)

        names = [f"{prefix}down_blocks.0"] + [f"{prefix}mid_blocks.{i}" for i in range(cfg.est_mid_blocks)] + [f"{prefix}up_blocks.0"]
        self.blocks = [dict(res=resnet(f"{n}.0"), tb=[tblock(f"{n}.1.{j}") for j in range(cfg.est_n_blocks)]) for n in names]
        if self.fused:
            for blk in self.blocks:   # fragment-ordered conv weights of the row-block resnet kernels (K zero-padded to whole groups)
                rs = blk["res"]
                for k in ("w1", "w2", "wr"):
                    w = rs[k]
                    kp = _round_up(w.shape[1], 128)
                    wp = torch.zeros(w.shape[0], kp, device=w.device, dtype=w.dtype)
                    wp[:, :w.shape[1]] = w
                    rs[k + "_p"] = ops.pack_skinny(wp)
        self.down_w, self.down_b = P.conv(f"{prefix}down_blocks.0.2.weight"), P.f32(f"{prefix}down_blocks.0.2.bias")
        self.up_w, self.up_b = P.conv(f"{prefix}up_blocks.0.2.weight"), P.f32(f"{prefix}up_blocks.0.2.bias")
        self.fin_w, self.fin_b = P.conv(f"{prefix}final_block.block.0.weight"), P.f32(f"{prefix}final_block.block.0.bias")
        self.fin_g, self.fin_be = P.f32(f"{prefix}final_block.block.2.weight"), P.f32(f"{prefix}final_block.block.2.bias")
        self.proj_w, self.proj_b = P.conv(f"{prefix}final_proj.weight"), P.f32(f"{prefix}final_proj.bias")
        self._load_time(P, sd, prefix, names)

    def time_table(self, t_values: List[float]) -> torch.Tensor:
        """(len(t), n_blocks*C) fp32: the per-resnet additive time term mlp(mish(time_mlp(sinus(t)))) for each step
        (flow/decoder.py:240-241, components/decoder.py:12-27,56).  Depends only on the step schedule."""
        key = tuple(round(float(t), 9) for t in t_values)
        if key in self._tcache:
            return self._tcache[key]
        cfg = self.cfg
        dim = cfg.est_in_channels
        half = dim // 2
        t = torch.tensor(t_values, dtype=torch.float32)
        e = math.log(10000) / (half - 1)
        e = torch.exp(torch.arange(half, dtype=torch.float32) * -e)
        e = 1000.0 * t.unsqueeze(1) * e.unsqueeze(0)
        se = torch.cat((e.sin(), e.cos()), dim=-1)
        n = len(t_values)
        npad = _round_up(n, 4)
        sin_in = torch.zeros(npad, dim)
        sin_in[:n] = se
        sin_in = sin_in.to(device=self.device, dtype=self.dtype)
        td = cfg.est_time_dim
        h1 = torch.empty(npad, td, device=self.device, dtype=self.dtype)
        h2 = torch.empty(npad, td, device=self.device, dtype=self.dtype)
        out = torch.empty(npad, self.tm_w.shape[0], device=self.device)
        ops.linear(sin_in, self.t1_w, bias=self.t1_b, act=ops.ACT_SILU, out_act=h1)
        ops.linear(h1, self.t2_w, bias=self.t2_b, act=ops.ACT_MISH, out_act=h2)
        ops.linear(h2, self.tm_w, bias=self.tm_b, out_f32=out)
        self._tcache[key] = out
        return out

    def _workspace(self, R, T):
        key = (R, T)
        if key in self._ws:
            return self._ws[key]
        cfg, dt, dev = self.cfg, self.dtype, self.device
        C, inner, ff = cfg.est_channels, cfg.est_inner, cfg.est_channels * cfg.est_ff_mult
        e = lambda *s, dtype=torch.float32: torch.empty(*s, device=dev, dtype=dtype)
        Tp = _round_up(T, 8)
        ws = dict(T=T, Tp=Tp, xin=e(R, T, cfg.est_in_channels, dtype=dt), c32a=e(R, T, C), c32b=e(R, T, C), h1=e(R, T, C, dtype=dt),
                  x32=e(R, T, C), xn=e(R, T, C, dtype=dt), qk=e(R, T, 2 * inner, dtype=dt),
                  vt=torch.zeros(R, cfg.est_heads, 64, Tp, device=dev, dtype=dt), ao=e(R, T, inner, dtype=dt),
                  ff=e(R, T, ff, dtype=dt), cat=e(R, T, 2 * C, dtype=dt), d=e(R, T, C, dtype=dt), v=e(R, T, cfg.output_size))
        self._ws[key] = ws
        return ws

    def _resnet(self, rs, ws, R, a_in, lda, cin, tadd):
        C, T = self.cfg.est_channels, ws["T"]
        rows = R * T
        if "w1_p" in rs and self.fused and cin in (256, 320, 512) and os.environ.get("CV_FLOW_FUSED_RESNET", "1") != "0":
            # two row-block launches: conv + LayerNorm + Mish + time term | conv + LayerNorm + Mish + 1x1 conv of the input
            p = L.ResblockParams()
            p.dtype, p.R, p.T, p.C, p.cin = L.TORCH_DT[self.dtype], R, T, C, cin
            p.a, p.lda = a_in.data_ptr(), lda
            p.w1_p, p.b1, p.g1, p.be1, p.tadd = rs["w1_p"].data_ptr(), rs["b1"].data_ptr(), rs["g1"].data_ptr(), rs["be1"].data_ptr(), tadd.data_ptr()
            p.h1, p.ldh1 = ws["h1"].data_ptr(), C
            p.w2_p, p.b2, p.g2, p.be2 = rs["w2_p"].data_ptr(), rs["b2"].data_ptr(), rs["g2"].data_ptr(), rs["be2"].data_ptr()
            p.wr_p, p.br = rs["wr_p"].data_ptr(), rs["br"].data_ptr()
            p.out, p.ldo, p.eps = ws["x32"].data_ptr(), C, 1e-5
            p.cus = int(getattr(self, "cu_budget", 0) or 0)
            ops._issue("cv_resblock_conv1", p)
            ops._issue("cv_resblock_conv2", p)
            return
        kw = dict(batch=R, a_bs=(T * lda, 0), lda=lda, a_rows=T)
        c32a, c32b = ws["c32a"], ws["c32b"]
        ops.gemm(a_in, rs["w1"], T, C, 3 * cin, cin=cin, tap_base=-2, tap_step=1, bias=rs["b1"], out_f32=c32a,
                 o32_bs=(T * C, 0), ldo32=C, **kw)
        ops.layernorm(c32a.view(rows, C), rs["g1"], rs["be1"], 1e-5, act=ops.ACT_MISH, add=tadd, rows_per_group=rows,
                      out_act=ws["h1"].view(rows, C))
        ops.gemm(ws["h1"], rs["w2"], T, C, 3 * C, batch=R, a_bs=(T * C, 0), lda=C, a_rows=T, cin=C, tap_base=-2, tap_step=1,
                 bias=rs["b2"], out_f32=c32a, o32_bs=(T * C, 0), ldo32=C)
        ops.layernorm(c32a.view(rows, C), rs["g2"], rs["be2"], 1e-5, act=ops.ACT_MISH, out_f32=c32b.view(rows, C))
        ops.gemm(a_in, rs["wr"], T, C, cin, bias=rs["br"], res=c32b, res_bs=(T * C, 0), ldres=C, out_f32=ws["x32"],
                 o32_bs=(T * C, 0), ldo32=C, **kw)

    def _tb_params(self, tb, ws, R):
        """cv_tblock_params with the HEAD fields of ``tb`` filled in."""
        inner = self.cfg.est_inner
        p = ops.tblock_params(ws["x32"], R, ws["T"], 1e-5, self.dtype)
        p.cus = int(getattr(self, "cu_budget", 0) or 0)   # CUs of the stream these launches go to (0 = all): tile-size choice only
        p.g1, p.b1n, p.wqkv_p = tb["g1"].data_ptr(), tb["b1"].data_ptr(), tb["wqkv_p"].data_ptr()
        p.qk, p.ldqk, p.vt, p.vt_ld = ws["qk"].data_ptr(), 2 * inner, ws["vt"].data_ptr(), ws["Tp"]
        return p

    def _tb_attention(self, ws, R, klen):
        cfg = self.cfg
        inner, H, T, Tp = cfg.est_inner, cfg.est_heads, ws["T"], ws["Tp"]
        ops.attention(ws["qk"], ws["qk"][:, :, inner:], ws["vt"], ws["ao"], B=R, H=H, Hkv=H, Tq=T, Tk=T, scale=cfg.est_head_dim ** -0.5,
                      q_bs=T * 2 * inner, ldq=2 * inner, k_bs=T * 2 * inner, ldk=2 * inner, vt_ld=Tp, o_bs=T * inner, ldo=inner, klen=klen)

    def _tb_tail(self, tb, ws, R, out_act=None, ldoa=0, next_tb=None):
        """Tail of ``tb``; with ``next_tb`` the same launch continues with that block's head (cv_tblock_tail_head)."""
        p = self._tb_params(next_tb if next_tb is not None else tb, ws, R)
        p.ao, p.ldao, p.wo_p, p.bo = ws["ao"].data_ptr(), self.cfg.est_inner, tb["wo_p"].data_ptr(), tb["bo"].data_ptr()
        p.g3, p.b3n = tb["g3"].data_ptr(), tb["b3"].data_ptr()
        p.w1_p, p.bf1, p.w2_p, p.bf2 = tb["wf1_p"].data_ptr(), tb["bf1"].data_ptr(), tb["wf2_p"].data_ptr(), tb["bf2"].data_ptr()
        if next_tb is not None:
            assert out_act is None
            ops.tblock_tail_head(p)
            return
        if out_act is not None:
            p.out_act, p.ldoa = out_act.data_ptr(), ldoa
        ops.tblock_tail(p)

    @property
    def fuse_tail_head(self):
        """Blocks j < n_tb - 1 of a group run their tail together with the next block's head (one launch, x not re-read)."""
        return bool(self.fused) and os.environ.get("CV_FLOW_FUSE_TAIL_HEAD", "1") != "0"

    def _tblock_group(self, tbs, ws, R, klen, out_act=None, ldoa=0):
        """The n_tb transformer blocks behind one resnet block; the last one also leaves the 16-bit copy ``out_act``."""
        if not (self.fused and all("wqkv_p" in tb for tb in tbs)):
            for j, tb in enumerate(tbs):
                last = j == len(tbs) - 1
                self._tblock(tb, ws, R, klen, out_act=out_act if last else None, ldoa=ldoa if last else 0)
            return
        fuse = self.fuse_tail_head
        for j, tb in enumerate(tbs):
            last = j == len(tbs) - 1
            if j == 0 or not fuse:
                ops.tblock_head(self._tb_params(tb, ws, R))
            self._tb_attention(ws, R, klen)
            if last:
                self._tb_tail(tb, ws, R, out_act=out_act, ldoa=ldoa)
            else:
                self._tb_tail(tb, ws, R, next_tb=tbs[j + 1] if fuse else None)

    def _tblock(self, tb, ws, R, klen, out_act=None, ldoa=0):
        cfg = self.cfg
        C, inner, ff, H = cfg.est_channels, cfg.est_inner, cfg.est_channels * cfg.est_ff_mult, cfg.est_heads
        T, Tp = ws["T"], ws["Tp"]
        rows = R * T
        if "wqkv_p" in tb and self.fused:
            # three launches: LN + [Q | K | V^T]  ->  flash attention  ->  to_out + residual + LN + FFN + residual
            ops.tblock_head(self._tb_params(tb, ws, R))
            self._tb_attention(ws, R, klen)
            self._tb_tail(tb, ws, R, out_act=out_act, ldoa=ldoa)
            return
        x2 = ws["x32"].view(rows, C)
        ops.layernorm(x2, tb["g1"], tb["b1"], 1e-5, out_act=ws["xn"].view(rows, C))
        # Q | K row-major in one GEMM; V^T directly as the swapped product Wv . xn^T (rows = head*64 + d, keys contiguous)
        ops.linear(ws["xn"].view(rows, C), tb["wqk"], out_act=ws["qk"].view(rows, 2 * inner))
        ops.gemm(tb["wv"], ws["xn"], inner, T, C, batch=R, lda=C, w_bs=(T * C, 0), ldw=C, out_act=ws["vt"],
                 oa_bs=(inner * Tp, 0), ldoa=Tp)
        ops.attention(ws["qk"], ws["qk"][:, :, inner:], ws["vt"], ws["ao"], B=R, H=H, Hkv=H, Tq=T, Tk=T, scale=cfg.est_head_dim ** -0.5,
                      q_bs=T * 2 * inner, ldq=2 * inner, k_bs=T * 2 * inner, ldk=2 * inner, vt_ld=Tp, o_bs=T * inner, ldo=inner, klen=klen)
        ops.linear(ws["ao"].view(rows, inner), tb["wo"], bias=tb["bo"], res=x2, out_f32=x2)
        ops.layernorm(x2, tb["g3"], tb["b3"], 1e-5, out_act=ws["xn"].view(rows, C))
        ops.linear(ws["xn"].view(rows, C), tb["wf1"], bias=tb["bf1"], act=ops.ACT_GELU, out_act=ws["ff"].view(rows, ff))
        if out_act is None:
            ops.linear(ws["ff"].view(rows, ff), tb["wf2"], bias=tb["bf2"], res=x2, out_f32=x2)
        else:
            ops.gemm(ws["ff"], tb["wf2"], rows, C, ff, lda=ff, bias=tb["bf2"], res=x2, ldres=C, out_f32=x2, ldo32=C,
                     out_act=out_act, ldoa=ldoa)

    def forward_cl(self, ws, R, tadd_row: torch.Tensor, klen=None):
        """ws['xin'] (R,T,320) -> ws['v'] (R,T,80) fp32.  tadd_row: (n_blocks*C,) time terms of this step."""
        cfg = self.cfg
        C, T = cfg.est_channels, ws["T"]
        nb = len(self.blocks)
        cat = ws["cat"]
        a_in, lda, cin = ws["xin"], cfg.est_in_channels, cfg.est_in_channels
        for bi, blk in enumerate(self.blocks):
            self._resnet(blk["res"], ws, R, a_in, lda, cin, tadd_row[bi * C:(bi + 1) * C])
            if bi == 0:
                # skip connection: 16-bit copy of the group's output into cat[:, :, C:2C] (hiddens.append, decoder.py:277)
                self._tblock_group(blk["tb"], ws, R, klen, out_act=cat[:, :, C:], ldoa=2 * C)
            elif bi == nb - 2:
                self._tblock_group(blk["tb"], ws, R, klen, out_act=cat, ldoa=2 * C)  # last mid block -> cat[:, :, 0:C]
            else:
                self._tblock_group(blk["tb"], ws, R, klen, out_act=ws["d"], ldoa=C)
            if bi == 0:
                # downsample slot = CausalConv1d k3 on the skip tensor (decoder.py:278)
                ops.gemm(cat[:, :, C:], self.down_w, T, C, 3 * C, batch=R, a_bs=(T * 2 * C, 0), lda=2 * C, a_rows=T, cin=C,
                         tap_base=-2, tap_step=1, bias=self.down_b, out_act=ws["d"], oa_bs=(T * C, 0), ldoa=C)
                a_in, lda, cin = ws["d"], C, C
            elif bi == nb - 2:
                a_in, lda, cin = cat, 2 * C, 2 * C
            else:
                a_in, lda, cin = ws["d"], C, C
        # upsample slot = CausalConv1d k3; final block; final_proj (decoder.py:331-334)
        ops.conv1d_cl(ws["d"], self.up_w, 3, pad_left=2, bias=self.up_b, out_act=ws["h1"])
        ops.conv1d_cl(ws["h1"], self.fin_w, 3, pad_left=2, bias=self.fin_b, out_f32=ws["c32a"])
        rows = R * T
        ops.layernorm(ws["c32a"].view(rows, C), self.fin_g, self.fin_be, 1e-5, act=ops.ACT_MISH, out_act=ws["h1"].view(rows, C))
        ops.linear(ws["h1"].view(rows, C), self.proj_w, bias=self.proj_b, out_f32=ws["v"].view(rows, cfg.output_size))
        return ws["v"]

    @torch.no_grad()
    def __call__(self, x, mask, mu, t, spks, cond, streaming=False):
        """Reference slot contract: x,mu,cond (R,80,T), mask (R,1,T) (all ones), t (R,), spks (R,80) -> (R,80,T) fp32."""
        R, Cm, T = x.shape
        dev = self.device
        ws = self._workspace(R, T)
        f = lambda a: a.to(dev, torch.float32).contiguous()
        xin = torch.cat([f(x), f(mu), f(spks).unsqueeze(-1).expand(-1, -1, T), f(cond)], dim=1).contiguous()  # (R,320,T)
        ops.to_channels_last(xin, ws["xin"])
        tt = self.time_table([float(t[0])])
        self.forward_cl(ws, R, tt[0])
        out = torch.empty(R, Cm, T, device=dev)
        ops.to_channels_first(ws["v"], out)
        return out


# =============================================================================== CFM + flow
class CausalConditionalCFM:
    def __init__(self, cfg: FlowConfig, estimator: ConditionalDecoder, device):
        self.cfg, self.estimator, self.device = cfg, estimator, device
        self.inference_cfg_rate = cfg.inference_cfg_rate
        # flow_matching.py:212-213: set_all_random_seed(0); rand_noise = randn([1, 80, 50*300]) — regenerated, never stored
        g_state = torch.get_rng_state()
        torch.manual_seed(0)
        self.rand_noise = torch.randn([1, 80, cfg.noise_len])
        torch.set_rng_state(g_state)
        self._noise_cl = self.rand_noise[0].t().contiguous().to(device)  # (15000, 80) channels-last
        self._graphs: Dict[tuple, tuple] = {}
        self.use_graph = False
        self.use_stage_abi = os.environ.get("CV_FLOW_STAGE_ABI", "1") != "0"   # solver graph built by cv_flow_euler_graph_create

    def schedule(self, n_timesteps):
        """t values fed to the estimator and the dt of every Euler step, with the reference's fp32 accumulation order
        (flow_matching.py:88,118-122,237-239)."""
        t_span = torch.linspace(0, 1, n_timesteps + 1, dtype=torch.float32)
        t_span = 1 - torch.cos(t_span * 0.5 * torch.pi)
        t, dt = t_span[0], t_span[1] - t_span[0]
        ts, dts = [], []
        for step in range(1, n_timesteps + 1):
            ts.append(float(t))
            dts.append(float(dt))
            t = t + dt
            if step < n_timesteps:
                dt = t_span[step + 1] - t
        return ts, dts

    def solve(self, x, mu, spks, cond, n_timesteps, klen=None):
        """x (B,T,80) fp32 state (updated in place); mu, cond (B,T,80); spks (B,80).  flow_matching.py:72-124."""
        est = self.estimator
        B, T, _ = x.shape
        R = 2 * B
        ws = est._workspace(R, T)
        ts, dts = self.schedule(n_timesteps)
        tt = est.time_table(ts)
        klen2 = None
        if klen is not None:   # persistent buffer: its address is baked into the captured graph, its contents change per call
            if not hasattr(self, "_klen2"):
                self._klen2 = {}
            klen2 = self._klen2.get(B)
            if klen2 is None:
                klen2 = self._klen2[B] = torch.zeros(2 * B, device=x.device, dtype=torch.int32)
            klen2.copy_(klen.repeat_interleave(2))

        def run():
            for i in range(n_timesteps):
                ops.est_pack(x, mu, spks, cond, ws["xin"])
                est.forward_cl(ws, R, tt[i], klen2)
                ops.cfm_update(x, ws["v"], dts[i], self.inference_cfg_rate)

        # stage-level ABI (cv_flow_euler_*): the library composes and captures the same launch sequence from a descriptor;
        # `run` stays as the eager path and as the cross-check of the C composition (tests/test_flow_gpu.py)
        stage_abi = self.use_stage_abi and getattr(est, "fused_all", False)   # (the v1 estimator, flow_v1.py, has no fused form)

        if not self.use_graph:
            run()
            return x
        key = (B, T, n_timesteps, x.data_ptr(), mu.data_ptr(), spks.data_ptr(), cond.data_ptr(),
               0 if klen2 is None else klen2.data_ptr(), int(getattr(est, "cu_budget", 0) or 0), est.fused, getattr(est, "fuse_tail_head", False))
        key = key + (stage_abi,)
        g = self._graphs.get(key)
        if g is None:
            run()  # warm every lazily-built table outside capture
            torch.cuda.synchronize()
            if stage_abi:
                desc, keep = self.solver_desc(x, mu, spks, cond, klen2, ws, tt, dts)
                g = ops.Graph.from_flow_solver(desc, keep)
            else:
                g = ops.Graph().capture(run)
            self._graphs[key] = g
            return x  # the eager warm-up already produced the result (x was advanced once)
        g.launch()
        return x

    def solver_desc(self, x, mu, spks, cond, klen2, ws, tt, dts):
        """cv_flow_solver_desc of this solve (include/cosyvoice_amd.h) + the host arrays it points to."""
        from . import _lib as L
        import ctypes as C
        est, cfg = self.estimator, self.cfg
        B, T, _ = x.shape
        d = L.FlowSolverDesc()
        d.dtype, d.B, d.T, d.Tp = L.TORCH_DT[est.dtype], B, T, ws["Tp"]
        d.C, d.inner, d.ff, d.heads = cfg.est_channels, cfg.est_inner, cfg.est_channels * cfg.est_ff_mult, cfg.est_heads
        d.in_ch, d.out_ch = cfg.est_in_channels, cfg.output_size
        d.n_blocks, d.n_steps, d.cus = len(est.blocks), len(dts), int(getattr(est, "cu_budget", 0) or 0)
        d.cfg_rate, d.eps = self.inference_cfg_rate, 1e-5
        blocks = (L.FlowBlock * len(est.blocks))()
        keep = [blocks]
        for bi, blk in enumerate(est.blocks):
            rs, r = blk["res"], blocks[bi].res
            for f, k in (("w1_p", "w1_p"), ("b1", "b1"), ("g1", "g1"), ("be1", "be1"), ("w2_p", "w2_p"), ("b2", "b2"), ("g2", "g2"),
                         ("be2", "be2"), ("wr_p", "wr_p"), ("br", "br")):
                setattr(r, f, rs[k].data_ptr())
            r.cin = rs["wr"].shape[1]
            tbs = (L.FlowTBlock * len(blk["tb"]))()
            keep.append(tbs)
            for j, tb in enumerate(blk["tb"]):
                for f, k in (("g1", "g1"), ("b1n", "b1"), ("wqkv_p", "wqkv_p"), ("wo_p", "wo_p"), ("bo", "bo"), ("g3", "g3"), ("b3n", "b3"),
                             ("w1_p", "wf1_p"), ("bf1", "bf1"), ("w2_p", "wf2_p"), ("bf2", "bf2")):
                    setattr(tbs[j], f, tb[k].data_ptr())
            blocks[bi].tb, blocks[bi].n_tb = tbs, len(blk["tb"])
            blocks[bi].fuse_tail_head = int(est.fuse_tail_head)
        d.blocks = blocks
        for f in ("down_w", "down_b", "up_w", "up_b", "fin_w", "fin_b", "fin_g", "fin_be", "proj_w", "proj_b"):
            setattr(d, f, getattr(est, f).data_ptr())
        tt_c = tt.contiguous()
        dts_c = (C.c_float * len(dts))(*dts)
        keep += [tt_c, dts_c]
        d.tadd, d.dts = tt_c.data_ptr(), dts_c
        d.x, d.mu, d.spks, d.cond = x.data_ptr(), mu.data_ptr(), spks.data_ptr(), cond.data_ptr()
        d.klen = None if klen2 is None else klen2.data_ptr()
        for f in ("xin", "h1", "x32", "qk", "vt", "ao", "cat", "d", "v", "c32a"):
            setattr(d, f, ws[f].data_ptr())
        return d, keep

    @torch.no_grad()
    def forward(self, mu, mask, n_timesteps, temperature=1.0, spks=None, cond=None):
        """Reference signature (flow_matching.py:215-240): mu, cond (1,80,T), spks (1,80) -> ((1,80,T) fp32, None)."""
        B, Cm, T = mu.shape
        dev = self.device
        mu_cl = mu.to(dev, torch.float32).transpose(1, 2).contiguous()
        cond_cl = cond.to(dev, torch.float32).transpose(1, 2).contiguous()
        x = (self._noise_cl[:T] * temperature).unsqueeze(0).repeat(B, 1, 1).contiguous()
        self.solve(x, mu_cl, spks.to(dev, torch.float32).contiguous(), cond_cl, n_timesteps)
        out = torch.empty(B, Cm, T, device=dev)
        ops.to_channels_first(x, out)
        return out, None

    __call__ = forward


class CausalMaskedDiffWithXvec:
    def __init__(self, cfg: Optional[FlowConfig] = None, dtype: torch.dtype = torch.bfloat16, device: str = "cuda"):
        self.cfg = cfg or FlowConfig.full()
        self.dtype, self.device = dtype, torch.device(device)
        self.fp16 = False
        self.input_frame_rate = self.cfg.input_frame_rate
        self.token_mel_ratio = self.cfg.token_mel_ratio
        self.pre_lookahead_len = self.cfg.pre_lookahead_len
        self.output_size = self.cfg.output_size
        self.encoder = UpsampleConformerEncoder(self.cfg, dtype, self.device)
        self.decoder = CausalConditionalCFM(self.cfg, ConditionalDecoder(self.cfg, dtype, self.device), self.device)
        self._loaded = False
        self._bufs: Dict[tuple, dict] = {}
        self._stream_caches: Dict[object, "EncoderStreamCache"] = {}   # cache_key of a streaming request -> its encoder cache

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    def half(self):
        return self

    def load_state_dict(self, sd, strict: bool = False):
        ops.drop_graphs(self.decoder._graphs)   # captured Euler loops hold raw pointers of the estimator weights replaced below
        sd = {k: v.detach().to("cpu") for k, v in sd.items()}
        P = _P(sd, self.dtype, self.device)
        self.emb_table = P.f32("input_embedding.weight")
        self.spk_w, self.spk_b = P.w("spk_embed_affine_layer.weight"), P.f32("spk_embed_affine_layer.bias")
        self.proj_w, self.proj_b = P.w("encoder_proj.weight"), P.f32("encoder_proj.bias")
        self.encoder.load(sd)
        self.decoder.estimator.load(sd)
        self._loaded = True
        return self

    def _buffers(self, B, N, D):
        key = (B, N, D)
        # every per-shape cache of the flow hangs off this one (encoder / estimator workspaces, position tables, the captured
        # Euler loops that hold their addresses): bounded together
        enc, est = self.encoder, self.decoder.estimator
        ops.bound_cache(self._bufs, key, enc._ws, enc._pos, est._ws, self.decoder._graphs, getattr(self.decoder, "_klen2", {}),
                        cap=getattr(self, "shape_cache_cap", None))
        if key not in self._bufs:
            dev = self.device
            Dp = _round_up(D, 8)
            self._bufs[key] = dict(idx=torch.empty(B, N, device=dev, dtype=torch.int32),
                                   emb_in=torch.zeros(B, D, device=dev), emb_n=torch.zeros(B, Dp, device=dev, dtype=self.dtype),
                                   spks=torch.empty(B, self.cfg.output_size, device=dev),
                                   cond=torch.zeros(B, 2 * N, self.cfg.output_size, device=dev),
                                   x=torch.empty(B, 2 * N, self.cfg.output_size, device=dev),
                                   mel=torch.empty(B, self.cfg.output_size, 2 * N, device=dev),
                                   klen_enc=torch.zeros(B, device=dev, dtype=torch.int32),
                                   klen_est=torch.zeros(B, device=dev, dtype=torch.int32))
        return self._bufs[key]

    @torch.no_grad()
    def inference_batch(self, tokens: torch.Tensor, prompt_tokens: torch.Tensor, prompt_feats: torch.Tensor,
                        embeddings: torch.Tensor, n_timesteps: int = 10, stream_cache=None):
        """B equal-shape utterances.  tokens (B,Ng) i32, prompt_tokens (B,Np), prompt_feats (B,Tp,80) with Tp = 2*Np,
        embeddings (B,D) -> mel (B,80,2*Ng) fp32.  Per utterance identical to ``inference`` (flow.py:258-319).
        ``stream_cache`` (EncoderStreamCache, B = 1): a streaming request's chunk call — the encoder only runs on the rows behind
        the cached chunk-causal prefix (UpsampleConformerEncoder.forward_tokens_cached)."""
        assert self._loaded
        cfg, dev = self.cfg, self.device
        B, Ng = tokens.shape
        Np = prompt_tokens.shape[1]
        Nv = Np + Ng                       # real tokens
        # length bucketing (off by default): run at the next multiple of ``length_bucket`` tokens with the tail masked, so a
        # stream of requests of arbitrary lengths re-uses a handful of captured estimator graphs instead of capturing one per
        # length (339 ms instead of 138 ms to the first chunk, tools/stream_latency.py).  Exact: the padded tail is invisible
        # to attention (klen), the convolutions only look left (the look-ahead conv sees zeros, as in the reference).
        lb = int(getattr(self, "length_bucket", 0) or 0)
        N = -(-Nv // lb) * lb if lb > 0 else Nv
        padded = lb > 0                    # with bucketing on, an exact fit also takes the masked path: one graph per bucket
        T = 2 * N
        D = embeddings.shape[1]
        bf = self._buffers(B, N, D)
        # speaker: F.normalize -> Linear (flow.py:286-287); RMSNorm kernel with scale 1/sqrt(D) == x / ||x||
        bf["emb_in"].copy_(embeddings.to(dev, torch.float32))
        ops.layernorm(bf["emb_in"], None, None, 1e-24 / D, rms=True, out_scale=1.0 / math.sqrt(D), out_act=bf["emb_n"][:, :D])
        ops.gemm(bf["emb_n"], self.spk_w, B, cfg.output_size, D, lda=bf["emb_n"].stride(0), bias=self.spk_b, out_f32=bf["spks"],
                 ldo32=cfg.output_size)
        # tokens -> embedding -> encoder -> encoder_proj = mu (flow.py:290-302)
        bf["idx"][:, :Np].copy_(prompt_tokens.to(dev, torch.int32))
        bf["idx"][:, Np:Nv].copy_(tokens.to(dev, torch.int32))
        klen_enc = klen_est = None
        if padded:
            bf["idx"][:, Nv:].fill_(-1)
            klen_enc, klen_est = bf["klen_enc"], bf["klen_est"]
            klen_enc.fill_(Nv)
            klen_est.fill_(2 * Nv)
        ews = self.encoder._workspace(B, N)
        ops.embedding(self.emb_table, bf["idx"].view(-1), ews["tok"].view(B * N, cfg.enc_dim))
        if stream_cache is not None and B == 1 and int(self.encoder.static_chunk_size) > 0 and N <= stream_cache.cap:
            xa = self.encoder.forward_tokens_cached(stream_cache, ews["tok"], N, Nv, klen=klen_enc)
        else:
            self.encoder.forward_tokens(ews["tok"], B, N, klen=klen_enc, n_valid=Nv if padded else None)
            xa = ews["b"]["xa"]
        ops.linear(xa.view(B * T, cfg.enc_dim), self.proj_w, bias=self.proj_b, out_f32=ews["mu"].view(B * T, cfg.output_size))
        # conditions: prompt mel then zeros (flow.py:305-307)
        Tp = prompt_feats.shape[1]
        bf["cond"].zero_()
        bf["cond"][:, :Tp].copy_(prompt_feats.to(dev, torch.float32))
        bf["x"].copy_(self.decoder._noise_cl[:T].unsqueeze(0).expand(B, -1, -1))
        self.decoder.solve(bf["x"], ews["mu"], bf["spks"], bf["cond"], n_timesteps, klen=klen_est)
        ops.to_channels_first(bf["x"], bf["mel"])
        return bf["mel"][:, :, Tp:2 * Nv]

    @torch.no_grad()
    def inference_ragged(self, tokens, prompt_tokens, prompt_feats, embeddings: torch.Tensor, n_timesteps: int = 10):
        """B utterances of DIFFERENT lengths in one batched pass: lists of tokens (Ng_b,), prompt_tokens (Np_b,), prompt_feats
        (2*Np_b, 80); embeddings (B, D).  Returns a list of mels (80, 2*Ng_b) fp32, each identical to its batch-1 result: rows
        are padded to the longest (rounded up to ``length_bucket``) and the tails masked exactly as in a length-bucketed call."""
        assert self._loaded
        cfg, dev = self.cfg, self.device
        B = len(tokens)
        nps = [int(t.numel()) for t in prompt_tokens]
        nvs = [nps[b] + int(tokens[b].numel()) for b in range(B)]
        lb = int(getattr(self, "length_bucket", 0) or 0) or 1
        N = -(-max(nvs) // lb) * lb
        T = 2 * N
        D = embeddings.shape[1]
        bf = self._buffers(B, N, D)
        bf["emb_in"].copy_(embeddings.to(dev, torch.float32))
        ops.layernorm(bf["emb_in"], None, None, 1e-24 / D, rms=True, out_scale=1.0 / math.sqrt(D), out_act=bf["emb_n"][:, :D])
        ops.gemm(bf["emb_n"], self.spk_w, B, cfg.output_size, D, lda=bf["emb_n"].stride(0), bias=self.spk_b, out_f32=bf["spks"],
                 ldo32=cfg.output_size)
        bf["idx"].fill_(-1)
        bf["cond"].zero_()
        for b in range(B):
            bf["idx"][b, :nps[b]].copy_(prompt_tokens[b].reshape(-1).to(dev, torch.int32))
            bf["idx"][b, nps[b]:nvs[b]].copy_(tokens[b].reshape(-1).to(dev, torch.int32))
            assert prompt_feats[b].shape[-2] == 2 * nps[b]
            bf["cond"][b, :2 * nps[b]].copy_(prompt_feats[b].reshape(-1, cfg.output_size).to(dev, torch.float32))
        klen_enc, klen_est = bf["klen_enc"], bf["klen_est"]
        klen_enc.copy_(torch.tensor(nvs, dtype=torch.int32))
        klen_est.copy_(torch.tensor([2 * n for n in nvs], dtype=torch.int32))
        ews = self.encoder._workspace(B, N)
        ops.embedding(self.emb_table, bf["idx"].view(-1), ews["tok"].view(B * N, cfg.enc_dim))
        self.encoder.forward_tokens(ews["tok"], B, N, klen=klen_enc, n_valid=nvs)
        ops.linear(ews["b"]["xa"].view(B * T, cfg.enc_dim), self.proj_w, bias=self.proj_b, out_f32=ews["mu"].view(B * T, cfg.output_size))
        bf["x"].copy_(self.decoder._noise_cl[:T].unsqueeze(0).expand(B, -1, -1))
        self.decoder.solve(bf["x"], ews["mu"], bf["spks"], bf["cond"], n_timesteps, klen=klen_est)
        ops.to_channels_first(bf["x"], bf["mel"])
        return [bf["mel"][b, :, 2 * nps[b]:2 * nvs[b]].clone() for b in range(B)]

    supports_stream_cache = True

    def drop_stream_cache(self, cache_key):
        """End of a streaming request: release its encoder cache."""
        self._stream_caches.pop(cache_key, None)

    @torch.no_grad()
    def inference(self, token, token_len, prompt_token, prompt_token_len, prompt_feat, prompt_feat_len, embedding,
                  flow_cache=None, sample_rate=24000, n_timesteps=10, begin=False, finalize=True, cache_key=None):
        """Reference signature (flow.py:260-272) -> (mel (1,80,T_g) float32, None).  ``cache_key`` (extra keyword, absent from the
        reference): identifies a streaming request across its chunk calls; its chunk-causal encoder state is then kept between the
        calls (CV_STREAM_ENC_CACHE=0 turns that off: every call re-encodes everything, as the reference does)."""
        assert token.shape[0] == 1
        r = self.token_mel_ratio
        if int(prompt_feat_len[0]) % r != 0:  # flow.py:279-283 (mutates the length tensors in place, as the reference)
            prompt_feat_len[0] -= prompt_feat_len[0] % r
            prompt_feat = prompt_feat[:, :int(prompt_feat_len[0]), :]
            prompt_token_len[0] = prompt_feat_len[0] // r
            prompt_token = prompt_token[:, :int(prompt_token_len[0])]
        ec = None
        if cache_key is not None and os.environ.get("CV_STREAM_ENC_CACHE", "1") != "0" and int(self.encoder.static_chunk_size) > 0:
            need = int(prompt_token.shape[1] + token.shape[1]) + 64
            ec = self._stream_caches.get(cache_key)
            if ec is None or ec.cap < need:     # first call of the request, or the request outgrew its cache: start over, larger
                while len(self._stream_caches) >= 8:
                    self._stream_caches.pop(next(iter(self._stream_caches)))
                ec = self.encoder.new_stream_cache(max(1024, _round_up(2 * need, 256)))
                self._stream_caches[cache_key] = ec
        mel = self.inference_batch(token, prompt_token, prompt_feat, embedding, n_timesteps, stream_cache=ec)
        return mel.float().clone(), None
