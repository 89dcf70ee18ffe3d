"""Qwen2LM_Phoneme_Src2 on MI355X — the LM of every recipe of the fork (/root/reference/cosyvoice/llm/llm.py:1450-1772,
examples/tts_vc/cosyvoice2/conf/cosyvoice_pho_tts.yaml:29-73).  It differs from Qwen2LM only in how the prefill sequence is
built: 4-factor phoneme embeddings -> ConformerEncoder (rel-pos self-attention, no cnn module, no macaron) ->
text_encoder_affine_layer -> one DecoderLayer (self-attention over the phonemes, source attention over the BPE text embeddings)
-> lm_input = [sos, speaker, fused phonemes, task_id, prompt speech]; the decode loop, KV caches, graph and sampler kernels are
Qwen2LM's (``generate_batch(lm_inputs=...)``), with the fork's ``non_random_ras_sampling`` as sampler (fallback_mode 1).

Kernels: the conformer layers are the flow encoder's (``flow.UpsampleConformerEncoder._layer``); the DecoderLayer's 16 heads
of hidden/16 (= 56) channels are zero-padded to the attention kernel's 64-wide heads when the weights are packed."""
import math
from types import SimpleNamespace
from typing import Dict, Optional

import torch

from . import ops
from .config import LlmConfig, PhonemeFrontConfig
from .flow import UpsampleConformerEncoder, _P, _round_up
from .llm import Qwen2LM


class _TextEncoder(UpsampleConformerEncoder):
    """transformer/encoder.py ConformerEncoder (input_layer 'linear', rel_pos_espnet, rel_selfattn, full attention)."""

    def __init__(self, pcfg: PhonemeFrontConfig, dtype, device):
        cfg = SimpleNamespace(enc_dim=pcfg.enc_dim, enc_heads=pcfg.enc_heads, enc_linear_units=pcfg.enc_linear_units,
                              enc_blocks=pcfg.enc_blocks, input_size=pcfg.input_size)
        super().__init__(cfg, dtype, device)
        assert pcfg.enc_dim // pcfg.enc_heads == 64, "the attention kernel is built for 64-wide heads"

    def load(self, sd, prefix="text_encoder."):
        self._invalidate()
        P = _P(sd, self.dtype, self.device)
        self.embed = self._load_embed(P, prefix + "embed")
        self.layers = [self._load_layer(P, sd, f"{prefix}encoders.{i}") for i in range(self.cfg.enc_blocks)]
        self.after_g, self.after_b = P.f32(prefix + "after_norm.weight"), P.f32(prefix + "after_norm.bias")

    def _workspace(self, R, T):
        key = (R, T)
        if key not in self._ws:
            cfg, dt, dev = self.cfg, self.dtype, self.device
            D, H, U = cfg.enc_dim, cfg.enc_heads, cfg.enc_linear_units
            e = lambda *s, dtype=torch.float32: torch.empty(*s, device=dev, dtype=dtype)
            Tp, ldb = _round_up(T, 8), _round_up(2 * T - 1, 4)
            self._ws[key] = dict(a=dict(T=T, Tp=Tp, ldb=ldb, xs=e(R, T, D), xn=e(R, T, D, dtype=dt), lin=e(R, T, D),
                                        q=e(R, T, 3 * D, dtype=dt), vt=torch.zeros(R, H, 64, Tp, device=dev, dtype=dt),
                                        bd=e(R, H, T, ldb), ao=e(R, T, D, dtype=dt), ff=e(R, T, U, dtype=dt), xa=e(R, T, D, dtype=dt)),
                                 x_in=torch.zeros(R, T, cfg.input_size, device=dev, dtype=dt))
        return self._ws[key]

    def forward(self, ws, R, T):
        """ws['x_in'] (R,T,input) operand dtype -> ws['a']['xa'] (R,T,D) operand dtype (after_norm output)."""
        wa = ws["a"]
        D = self.cfg.enc_dim
        self._embed(self.embed, ws["x_in"].view(R * T, -1), wa, R)
        pos = self._pos_proj(self.layers, "a", T)
        for i, l in enumerate(self.layers):
            self._layer(l, wa, R, pos[i], 0, None)
        ops.layernorm(wa["xs"].view(R * T, D), self.after_g, self.after_b, 1e-5, out_act=wa["xa"].view(R * T, D))
        return wa["xa"]


class _PaddedMHA:
    """MultiHeadedAttention(n_head, n_feat) (transformer/attention.py:36-135) with d_k = n_feat / n_head <= 64 zero-padded
    to the 64-wide heads of cv_attention: Wq/Wk/Wv rows and Wout columns are regrouped per head with zero fill, so scores
    and outputs are unchanged (scale stays 1/sqrt(d_k)); the value bias is folded into linear_out's bias (softmax rows sum to 1)."""

    def __init__(self, sd, name, heads, dtype, device):
        f = lambda k: sd[f"{name}.{k}"].detach().float()
        Hd = f("linear_q.weight").shape[0]
        dk = Hd // heads
        assert dk <= 64 and Hd % heads == 0
        self.heads, self.dk, self.Hd = heads, dk, Hd

        def pad_rows(w, b):
            wp = torch.zeros(heads * 64, w.shape[1])
            bp = torch.zeros(heads * 64)
            for h in range(heads):
                wp[h * 64:h * 64 + dk] = w[h * dk:(h + 1) * dk]
                if b is not None:
                    bp[h * 64:h * 64 + dk] = b[h * dk:(h + 1) * dk]
            return wp, bp

        wq, bq = pad_rows(f("linear_q.weight"), f("linear_q.bias"))
        wk, bk = pad_rows(f("linear_k.weight"), f("linear_k.bias") if f"{name}.linear_k.bias" in sd else None)
        wv, _ = pad_rows(f("linear_v.weight"), None)
        wo = f("linear_out.weight")
        wop = torch.zeros(Hd, heads * 64)
        for h in range(heads):
            wop[:, h * 64:h * 64 + dk] = wo[:, h * dk:(h + 1) * dk]
        bo = f("linear_out.bias") + wo @ f("linear_v.bias")
        to = lambda t: t.to(device=device, dtype=dtype).contiguous()
        f32 = lambda t: t.to(device=device, dtype=torch.float32).contiguous()
        self.wq, self.bq, self.wk, self.bk, self.wv, self.wo, self.bo = to(wq), f32(bq), to(wk), f32(bk), to(wv), to(wop), f32(bo)

    def __call__(self, q_in, kv_in, ws, tag, res, out):
        """q_in (Tq,Hd), kv_in (Tk,Hd) operand dtype; out (Tq,Hd) fp32 = res + attention output."""
        Tq, Tk, H, W = q_in.shape[0], kv_in.shape[0], self.heads, self.heads * 64
        Tkp = _round_up(Tk, 8)
        q, k, ao = ws[f"{tag}_q"], ws[f"{tag}_k"], ws[f"{tag}_ao"]
        vt = ws[f"{tag}_vt"]
        ops.linear(q_in, self.wq, bias=self.bq, out_act=q)
        ops.linear(kv_in, self.wk, bias=self.bk, out_act=k)
        ops.gemm(self.wv, kv_in, W, Tk, self.Hd, lda=self.Hd, ldw=kv_in.stride(0), out_act=vt, ldoa=Tkp)   # V^T = Wv . kv^T
        ops.attention(q, k, vt, ao, B=1, H=H, Hkv=H, Tq=Tq, Tk=Tk, scale=1.0 / math.sqrt(self.dk), q_bs=Tq * W, ldq=W,
                      k_bs=Tk * W, ldk=W, vt_ld=Tkp, o_bs=Tq * W, ldo=W)
        ops.linear(ao, self.wo, bias=self.bo, res=res, out_f32=out)


class Qwen2LM_Phoneme_Src2:
    def __init__(self, lcfg: Optional[LlmConfig] = None, pcfg: Optional[PhonemeFrontConfig] = None, dtype: torch.dtype = torch.bfloat16,
                 device: str = "cuda", **lm_kwargs):
        self.lcfg, self.pcfg = lcfg or LlmConfig.full(), pcfg or PhonemeFrontConfig.full()
        self.dtype, self.device = dtype, torch.device(device)
        self.lm = Qwen2LM(self.lcfg, dtype=dtype, device=device, **lm_kwargs)
        pc = self.pcfg
        self.lm.top_p, self.lm.top_k, self.lm.win_size, self.lm.tau_r = pc.top_p, pc.top_k, pc.win_size, pc.tau_r
        self.lm.fallback_mode, self.lm.top_p2, self.lm.top_k2 = 1, pc.top_p + 0.15, pc.top_k * pc.expand_scale   # common.py:116-123
        self.encoder = _TextEncoder(pc, dtype, self.device)
        self.speech_token_size = self.lcfg.speech_token_size
        self.fp16 = False
        self._ws: Dict[tuple, dict] = {}
        self._loaded = False

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    def half(self):
        return self

    def load_state_dict(self, sd, strict: bool = False):
        pc, dt, dev = self.pcfg, self.dtype, self.device
        self.lm.load_state_dict(sd)
        P = _P(sd, dt, dev)
        self.pho_tables = [P.f32(f"text_embedding.{i}.weight") for i in range(4)]
        self.encoder.load(sd)
        self.aff_w, self.aff_b = P.w("text_encoder_affine_layer.weight"), P.f32("text_encoder_affine_layer.bias")
        n = "src_attention.0"
        self.self_attn = _PaddedMHA(sd, f"{n}.self_attn", pc.src_heads, dt, dev)
        self.src_attn = _PaddedMHA(sd, f"{n}.src_attn", pc.src_heads, dt, dev)
        self.norms = [(P.f32(f"{n}.norm{i}.weight"), P.f32(f"{n}.norm{i}.bias")) for i in (1, 2, 3)]
        self.ff_w1, self.ff_b1 = P.w(f"{n}.feed_forward.w_1.weight"), P.f32(f"{n}.feed_forward.w_1.bias")
        self.ff_w2, self.ff_b2 = P.w(f"{n}.feed_forward.w_2.weight"), P.f32(f"{n}.feed_forward.w_2.bias")
        self.spk_w, self.spk_b = P.w("spk_embed_affine_layer.weight"), P.f32("spk_embed_affine_layer.bias")
        self._loaded = True
        return self

    def _workspace(self, P_, L):
        key = (P_, L)
        ops.bound_cache(self._ws, key, self.encoder._ws, self.encoder._pos)
        if key not in self._ws:
            dt, dev, H = self.dtype, self.device, self.lcfg.hidden_size
            W = self.pcfg.src_heads * 64
            e = lambda *s, dtype=torch.float32: torch.empty(*s, device=dev, dtype=dtype)
            z = lambda *s, dtype=torch.float32: torch.zeros(*s, device=dev, dtype=dtype)
            ws = dict(x=e(P_, H), xn=e(P_, H, dtype=dt), x2=e(P_, H), x3=e(P_, H), text=e(L, H, dtype=dt),
                      ff=e(P_, self.pcfg.src_linear_units, dtype=dt), idx=torch.empty(4, P_, device=dev, dtype=torch.int32),
                      tidx=torch.empty(L, device=dev, dtype=torch.int32))
            for tag, Tk in (("sa", P_), ("ca", L)):
                ws[f"{tag}_q"], ws[f"{tag}_k"], ws[f"{tag}_ao"] = e(P_, W, dtype=dt), e(Tk, W, dtype=dt), e(P_, W, dtype=dt)
                ws[f"{tag}_vt"] = z(W, _round_up(Tk, 8), dtype=dt)
            self._ws[key] = ws
        return self._ws[key]

    @torch.no_grad()
    def lm_input(self, text, pho, prompt_text, prompt_pho, prompt_speech_token, embedding) -> torch.Tensor:
        """llm.py:1700-1745 -> (L, hidden) fp32 prefill embedding sequence [sos, spk?, fused phonemes, task_id, prompt speech]."""
        assert self._loaded
        pc, lc, dt, dev = self.pcfg, self.lcfg, self.dtype, self.device
        H = lc.hidden_size
        text = torch.cat([prompt_text.reshape(-1), text.reshape(-1)]).to(dev, torch.int32)
        pho = torch.cat([prompt_pho.reshape(-1, 4), pho.reshape(-1, 4)], dim=0).to(dev, torch.int32)
        P_, L = pho.shape[0], text.numel()
        ws = self._workspace(P_, L)
        ews = self.encoder._workspace(1, P_)
        # 4-factor phoneme embedding, concatenated along channels; the prosody factor is zeroed unless use_frontend_prsd (:1712-1716)
        ws["idx"].copy_(pho.t())
        if not pc.use_frontend_prsd:
            ws["idx"][3].fill_(-1)
        off = 0
        xin = ews["x_in"].view(P_, -1)
        for i, tab in enumerate(self.pho_tables):
            d = tab.shape[1]
            ops.embedding(tab, ws["idx"][i], xin[:, off:off + d])
            off += d
        enc = self.encoder.forward(ews, 1, P_).view(P_, pc.enc_dim)
        ops.linear(enc, self.aff_w, bias=self.aff_b, out_f32=ws["x"])                       # text_encoder_affine_layer (:1538)
        ws["tidx"].copy_(text)
        ops.embedding(self.lm.embed_tokens, ws["tidx"], ws["text"])                          # BPE embeddings = memory (:1722)
        # DecoderLayer, normalize_before (decoder_layer.py:91-127): self-attn over the phonemes, src-attn over the text, ReLU FFN
        x, x2, x3, xn = ws["x"], ws["x2"], ws["x3"], ws["xn"]
        ops.layernorm(x, self.norms[0][0], self.norms[0][1], 1e-5, out_act=xn)
        self.self_attn(xn, xn, ws, "sa", x, x2)
        ops.layernorm(x2, self.norms[1][0], self.norms[1][1], 1e-5, out_act=xn)
        self.src_attn(xn, ws["text"], ws, "ca", x2, x3)
        ops.layernorm(x3, self.norms[2][0], self.norms[2][1], 1e-5, out_act=xn)
        ops.linear(xn, self.ff_w1, bias=self.ff_b1, act=ops.ACT_LEAKY, act_slope=0.0, out_act=ws["ff"])
        ops.linear(ws["ff"], self.ff_w2, bias=self.ff_b2, res=x3, out_f32=x3)
        # assemble [sos, spk, fused phonemes, task_id, prompt speech] (:1735-1745)
        has_spk = embedding is not None and embedding.shape[0] != 0
        ps = prompt_speech_token.reshape(-1).to(dev, torch.int32)
        n_spk = 1 if has_spk else 0
        Ltot = 1 + n_spk + P_ + 1 + ps.numel()
        out = torch.zeros(Ltot, H, device=dev)
        idx = torch.full((2, Ltot), -2, device=dev, dtype=torch.int32)
        idx[0, 0] = self.lm.sos_eos
        idx[0, 1 + n_spk + P_] = self.lm.task_id
        idx[1, 2 + n_spk + P_:] = ps
        ops.embedding(self.lm.llm_embedding, idx[0], out)
        ops.embedding(self.lm.speech_embedding, idx[1], out)
        if has_spk:
            D = embedding.shape[1]
            e_in = embedding.to(dev, torch.float32).contiguous()
            e_n = torch.zeros(1, _round_up(D, 8), device=dev, dtype=dt)
            # F.normalize -> Linear (:1727-1729): RMSNorm kernel with scale 1/sqrt(D) == x / ||x||
            ops.layernorm(e_in, None, None, 1e-24 / D, rms=True, out_scale=1.0 / math.sqrt(D), out_act=e_n[:, :D])
            ops.gemm(e_n, self.spk_w, 1, H, D, lda=e_n.stride(0), bias=self.spk_b, out_f32=out[1:2], ldo32=H)
        out[1 + n_spk:1 + n_spk + P_].copy_(x3)
        return out

    @torch.no_grad()
    def inference(self, text, text_len, prompt_text, prompt_text_len, prompt_speech_token, prompt_speech_token_len, embedding,
                  sampling: int = 25, max_token_text_ratio: float = 20, min_token_text_ratio: float = 2):
        """Reference signature (llm.py:1687-1699): text = (bpe ids (1,L), phoneme factors (1,P,4)), likewise prompt_text and
        the length tuples.  Generator of python ints."""
        (t, pho), (pt, ppho) = text, prompt_text
        x = self.lm_input(t, pho, pt, ppho, prompt_speech_token, embedding)
        tl, ptl = text_len[0], prompt_text_len[0]
        yield from self.lm.inference(text=t, text_len=tl, prompt_text=pt, prompt_text_len=ptl, prompt_speech_token=prompt_speech_token,
                                     prompt_speech_token_len=prompt_speech_token_len, embedding=embedding, sampling=sampling,
                                     max_token_text_ratio=max_token_text_ratio, min_token_text_ratio=min_token_text_ratio, lm_input=x)
