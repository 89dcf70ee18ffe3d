"""CosyVoice-v1 ``TransformerLM`` on MI355X (/root/reference/cosyvoice/llm/llm.py:41-237):
text embedding -> ConformerEncoder (rel-pos, causal: static_chunk_size 1) -> affine -> [sos, speaker, text, task_id, prompt speech]
-> 14-layer TransformerEncoder with relative-position attention -> llm_decoder -> non_random_ras_sampling.

Decode = prefill + K/V-cached steps, as the reference's ``forward_chunk`` loop with its attention cache (llm.py:220-237,
transformer/encoder.py:185-274): the prompt runs once through the causal stack (the flow encoder's HIP kernels: fused QKV GEMM,
rel-pos bias GEMM, flash attention with bias + causal mask, FFN; length rounded up to a bucket, the padded tail is invisible to
a causal model) and leaves every layer's keys / values in per-layer caches; every new token then takes one cached step
(``_CausalStack.decode_step``): weight-streaming skinny GEMMs (``cv_skinny_gemm``; ReLU FFN = its mode 3), ``cv_relpos_append``,
the relative-position bias of the new query against the cached keys as one small GEMM over a reversed distance table, and
``cv_attention`` with that bias over t + 1 keys.  The step's 129 launches are recorded once (``ops.Recorder``) and re-issued
with the few t-dependent fields patched: 1.03 ms per token at full size (14 layers x 1024, context 313 -> 563), 3.6x the
full-recompute path it is checked against (``incremental = False``: the whole causal pass per step, the cached step being its
last row; both match the reference's own cached loop, oracle/llm_v1.py 3e-6).  Sampling is ``cv_sample_ras`` in its
non-random-RAS mode."""
import ctypes as C
import math
from types import SimpleNamespace
from typing import Generator, List, Optional

import torch

from . import _lib as L
from . import ops
from .config import TransformerLMConfig
from .flow import UpsampleConformerEncoder, _P, _round_up
from .llm_phoneme import _TextEncoder


class _CausalStack(UpsampleConformerEncoder):
    """TransformerEncoder (transformer/encoder.py:276-386): LegacyLinearNoSubsampling + N TransformerEncoderLayer + after_norm."""

    def __init__(self, cfg: TransformerLMConfig, dtype, device):
        super().__init__(SimpleNamespace(enc_dim=cfg.llm_dim, enc_heads=cfg.llm_heads, enc_linear_units=cfg.llm_linear_units,
                                         enc_blocks=cfg.llm_blocks, input_size=cfg.llm_dim), dtype, device)
        assert cfg.llm_dim // cfg.llm_heads == 64, "the attention kernel is built for 64-wide heads"

    def load(self, sd, prefix="llm."):
        self._invalidate()
        P = _P(sd, self.dtype, self.device)
        self.embed = self._load_embed(P, prefix + "embed")
        self.layers = [self._load_layer(P, sd, f"{prefix}encoders.{i}", "norm1", "norm2") for i in range(self.cfg.enc_blocks)]
        self.after_g, self.after_b = P.f32(prefix + "after_norm.weight"), P.f32(prefix + "after_norm.bias")

    _workspace = _TextEncoder._workspace

    def forward(self, seq_act, T, fill_cache=0):
        """seq_act (T, D) operand dtype (rows >= the live length hold anything finite) -> fp32 (T, D) after_norm output.
        ``fill_cache`` = n > 0 also stores every layer's keys / values of rows [0, n) in the decode caches."""
        D = self.cfg.enc_dim
        ws = self._workspace(1, T)
        wa = ws["a"]
        # LegacyLinearNoSubsampling: Linear -> LayerNorm(1e-5) -> ReLU, then x * sqrt(d) (subsampling.py:352-372, embedding.py:257-270)
        ops.linear(seq_act, self.embed["w"], bias=self.embed["b"], out_f32=wa["lin"].view(T, D))
        ops.layernorm(wa["lin"].view(T, D), self.embed["g"], self.embed["beta"], 1e-5, act=ops.ACT_LEAKY, out_scale=math.sqrt(D),
                      out_f32=wa["xs"].view(T, D), out_act=wa["xa"].view(T, D))
        pos = self._pos_proj(self.layers, "a", T)
        relu = dict(act=ops.ACT_LEAKY, act_slope=0.0)
        for i, l in enumerate(self.layers):
            self._layer(l, wa, 1, pos[i], 0, None, act=relu, causal=True)
            if fill_cache:
                n = fill_cache
                self.kc[i][:n].copy_(wa["q"][0, :n, 2 * D:])
                self.vtc[i][:, :, :n].copy_(wa["vt"][0, :, :, :n])
        ops.layernorm(wa["xs"].view(T, D), self.after_g, self.after_b, 1e-5, out_f32=wa["lin"].view(T, D))
        return wa["lin"].view(T, D)

    # ---- cached decode step (TransformerEncoder.forward_chunk with att_cache, transformer/encoder.py:185-274) ----
    def build_decoder(self, max_len, seq, h_out):
        """Per-layer K (max_len, D) / V^T (H, 64, max_len) caches, the projected relative-position table of every layer for
        the distances 0 .. max_len-1 a new token sees (stored reversed, so that the rows a step needs are contiguous and in
        key order), and the recorded launch sequence of one decode step: input layer on row t of ``seq``, 14 x [LayerNorm,
        one weight-streaming skinny GEMM for [q+u | q+v | k | v], cv_relpos_append (16-bit queries, K row t, V^T column t),
        rel-pos bias GEMM over the t + 1 cached keys, attention with that bias, out projection added in place, LayerNorm,
        FFN (ReLU fused into the first skinny GEMM, residual into the second)], after_norm into ``h_out`` row 0.  Only the
        pointers / lengths that depend on t are patched between steps."""
        cfg, dt, dev = self.cfg, self.dtype, self.device
        D, H, U = cfg.enc_dim, cfg.enc_heads, cfg.enc_linear_units
        es = torch.empty(0, dtype=dt).element_size()
        Tpm, ldbm = _round_up(max_len, 8), _round_up(max_len, 4)
        z = lambda *sh, dtype=dt: torch.zeros(*sh, device=dev, dtype=dtype)
        self.dec_max = max_len
        self.kc = [z(max_len, D) for _ in self.layers]
        self.vtc = [z(H, 64, Tpm) for _ in self.layers]
        # EspnetRelPositionalEncoding (embedding.py:220-294): the row of key j for query t is pe(t - j); prev[r] = pe(max_len-1-r)
        dist = torch.arange(max_len - 1, -1, -1, dtype=torch.float32).unsqueeze(1)
        div = torch.exp(torch.arange(0, D, 2, dtype=torch.float32) * -(math.log(10000.0) / D))
        pe = torch.zeros(max_len, D)
        pe[:, 0::2], pe[:, 1::2] = torch.sin(dist * div), torch.cos(dist * div)
        pe = pe.to(device=dev, dtype=dt).contiguous()
        self.prev = []
        for l in self.layers:
            pr = z(max_len, D)
            ops.linear(pe, l["wpos"], out_act=pr)
            self.prev.append(pr)
        # A operands of the skinny GEMMs are 16-row blocks (rows >= 1 stay zero); weights packed once as MFMA fragment streams
        b = dict(xs=z(1, D, dtype=torch.float32), lin=z(1, D, dtype=torch.float32), xn=z(16, D), qkv=z(1, 4 * D, dtype=torch.float32),
                 qq=z(1, 2 * D), ao=z(16, D), ff=z(16, U), bd=z(H, ldbm, dtype=torch.float32))
        self.dbuf = b
        self.dw = []
        for l in self.layers:
            wqkv = torch.cat([l["wqqk"], l["wv"]], 0).contiguous()                      # (4 D, D): [q+u | q+v | k | v]
            bqkv = torch.cat([l["bqqk"], torch.zeros(D, device=dev)], 0).contiguous()   # the value bias lives in bout
            self.dw.append(dict(qkv=ops.pack_skinny(wqkv), bqkv=bqkv, out=ops.pack_skinny(l["wout"]), w1=ops.pack_skinny(l["w1"]),
                                w2=ops.pack_skinny(l["w2"])))
        self.embed_p = ops.pack_skinny(self.embed["w"])
        scale = 1.0 / math.sqrt(D // H)
        patches = []          # (params struct, field, base address, bytes per step of t)
        self._len_fields, self._t_args = [], []
        cdt = L.TORCH_DT[dt]
        vp = lambda t_: C.c_void_p(t_.data_ptr())
        rec = ops.Recorder()
        with rec:
            # LegacyLinearNoSubsampling on row t: Linear -> LayerNorm -> ReLU, x sqrt(D)
            ops.skinny_gemm(seq, self.embed_p, 1, D, D, bias=self.embed["b"], out_f32=b["lin"], ldo=D)
            patches.append((rec.calls[-1][1], "A", seq.data_ptr(), D * es))
            ops.layernorm(b["lin"], self.embed["g"], self.embed["beta"], 1e-5, act=ops.ACT_LEAKY, out_scale=math.sqrt(D), out_f32=b["xs"])
            for i, l in enumerate(self.layers):
                w = self.dw[i]
                ops.layernorm(b["xs"], l["g_mha"], l["b_mha"], 1e-12, out_act=b["xn"][:1])
                ops.skinny_gemm(b["xn"], w["qkv"], 1, 4 * D, D, bias=w["bqkv"], out_f32=b["qkv"], ldo=4 * D)
                args = [vp(b["qkv"]), vp(b["qq"]), vp(self.kc[i]), vp(self.vtc[i]), cdt, D, 0, Tpm]
                rec.add_raw("cv_relpos_append", args)
                self._t_args.append(args)
                ops.gemm(b["qq"][:, D:], self.prev[i], 1, 1, 64, batch=H, a_bs=(64, 0), lda=2 * D, w_bs=(64, 0), ldw=D,
                         out_scale=scale, out_f32=b["bd"], o32_bs=(ldbm, 0), ldo32=ldbm)
                patches.append((rec.calls[-1][1], "W", self.prev[i].data_ptr() + (max_len - 1) * D * es, -D * es))
                self._len_fields.append((rec.calls[-1][1], "N"))
                ops.attention(b["qq"], self.kc[i], self.vtc[i], b["ao"], B=1, H=H, Hkv=H, Tq=1, Tk=1, scale=scale, q_bs=2 * D, ldq=2 * D,
                              k_bs=max_len * D, ldk=D, vt_ld=Tpm, o_bs=D, ldo=D, bias=b["bd"], bias_bs=0, bias_hs=ldbm, bias_ld=ldbm)
                self._len_fields.append((rec.calls[-1][1], "Tk"))
                ops.skinny_gemm(b["ao"], w["out"], 1, D, D, bias=l["bout"], mode=1, out_f32=b["xs"], ldo=D)
                ops.layernorm(b["xs"], l["g_ff"], l["b_ff"], 1e-12, out_act=b["xn"][:1])
                ops.skinny_gemm(b["xn"], w["w1"], 1, U, D, bias=l["b1"], mode=3, out_act=b["ff"], ldoa=U)
                ops.skinny_gemm(b["ff"], w["w2"], 1, D, U, bias=l["b2"], mode=1, out_f32=b["xs"], ldo=D)
            ops.layernorm(b["xs"], self.after_g, self.after_b, 1e-5, out_act=h_out[:1])
        self._dec, self._patches = rec, patches

    def decode_step(self, t):
        """Row t of the sequence (its embedding already in ``seq[t]``) through the stack against the cached rows [0, t):
        after_norm output -> ``h_out[0]``; row t's keys / values join the caches."""
        assert 0 < t < self.dec_max
        for p, field, base, step in self._patches:
            setattr(p, field, base + t * step)
        for p, field in self._len_fields:
            setattr(p, field, t + 1)
        for a in self._t_args:
            a[6] = t
        self._dec.replay()


class _CausalTextEncoder(_TextEncoder):
    def forward(self, ws, R, T):
        wa = ws["a"]
        D = self.cfg.enc_dim
        self._embed(self.embed, ws["x_in"].view(R * T, -1), wa, R)
        pos = self._pos_proj(self.layers, "a", T)
        for i, l in enumerate(self.layers):
            self._layer(l, wa, R, pos[i], 1, None)      # static / decoding chunk size 1: attend to self and the past (llm.py:93)
        ops.layernorm(wa["xs"].view(R * T, D), self.after_g, self.after_b, 1e-5, out_act=wa["xa"].view(R * T, D))
        return wa["xa"]


class TransformerLM:
    def __init__(self, cfg: Optional[TransformerLMConfig] = None, dtype: torch.dtype = torch.float16, device: str = "cuda",
                 max_len: int = 2048, bucket: int = 64):
        if not torch.cuda.is_available():
            raise RuntimeError("cosyvoice_amd needs an MI355X (no CPU fallback)")
        self.cfg = cfg or TransformerLMConfig.full()
        self.dtype, self.device = dtype, torch.device(device)
        self.max_len, self.bucket = max_len, bucket
        c = self.cfg
        self.text_encoder = _CausalTextEncoder(SimpleNamespace(enc_dim=c.enc_dim, enc_heads=c.enc_heads, enc_linear_units=c.enc_linear_units,
                                                               enc_blocks=c.enc_blocks, input_size=c.text_encoder_input_size), dtype, self.device)
        self.stack = _CausalStack(c, dtype, self.device)
        self.speech_token_size = c.speech_token_size
        self.sos_eos, self.task_id = 0, 1
        self.fp16 = False
        self.seed = 0
        self._loaded = False

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    def half(self):
        return self

    def load_state_dict(self, sd, strict: bool = False):
        c, dt, dev = self.cfg, self.dtype, self.device
        P = _P(sd, dt, dev)
        self.text_table = P.f32("text_embedding.weight")
        self.text_encoder.load(sd)
        self.stack.load(sd)
        self.aff_w, self.aff_b = P.w("text_encoder_affine_layer.weight"), P.f32("text_encoder_affine_layer.bias")
        self.llm_embedding, self.speech_embedding = P.f32("llm_embedding.weight"), P.f32("speech_embedding.weight")
        self.spk_w, self.spk_b = P.w("spk_embed_affine_layer.weight"), P.f32("spk_embed_affine_layer.bias")
        V = c.speech_token_size + 1
        self.Vpad = _round_up(V, 4)
        wd = torch.zeros(self.Vpad, c.llm_dim)
        wd[:V] = sd["llm_decoder.weight"].float()
        bd = torch.zeros(self.Vpad)
        bd[:V] = sd["llm_decoder.bias"].float()
        self.dec_w, self.dec_b = wd.to(device=dev, dtype=dt).contiguous(), bd.to(dev)
        z = lambda *s, dtype=torch.float32: torch.zeros(*s, device=dev, dtype=dtype)
        self.seq = z(self.max_len + 16, c.llm_dim, dtype=dt)      # lm_input followed by the emitted speech embeddings, operand dtype
        self.dec_wp = ops.pack_skinny(self.dec_w)                  # (+ 16 spare rows: a skinny GEMM reads a 16-row A block)
        self.st = dict(x=z(16, c.llm_dim), h=z(16, c.llm_dim, dtype=dt), logits=z(16, self.Vpad), pos=z(16, dtype=torch.int32),
                       step=z(16, dtype=torch.int32), n_emitted=z(16, dtype=torch.int32), finished=z(16, dtype=torch.int32),
                       min_len=z(16, dtype=torch.int32), max_len=z(16, dtype=torch.int32),
                       out_tokens=z(16, self.max_len, dtype=torch.int32), uniforms=z(16, 101, 2), nonce=z(2, dtype=torch.int64))
        self.stack.build_decoder(self.max_len, self.seq, self.st["h"])
        self.incremental = True      # False: recompute the full causal pass every step (the cross-check of the cached path)
        self._cached = 0             # rows of the current request whose keys / values are in the caches
        self.n_decode_steps = 0      # cached steps taken since load (diagnostics / tests)
        self._loaded = True
        return self

    @torch.no_grad()
    def lm_input(self, text, prompt_text, prompt_speech_token, embedding) -> torch.Tensor:
        """llm.py:188-213 -> (L, llm_dim) fp32: [sos, speaker?, encoded text, task_id, prompt speech embeddings]."""
        assert self._loaded
        c, dt, dev = self.cfg, self.dtype, self.device
        ids = torch.cat([prompt_text.reshape(-1), text.reshape(-1)]).to(dev, torch.int32)
        Lt = ids.numel()
        ops.bound_cache(self.text_encoder._ws, (1, Lt), self.text_encoder._pos)      # one workspace / position table per text length
        ews = self.text_encoder._workspace(1, Lt)
        ops.embedding(self.text_table, ids, ews["x_in"].view(Lt, -1))
        enc = self.text_encoder.forward(ews, 1, Lt).view(Lt, c.enc_dim)
        has_spk = embedding is not None and embedding.shape[0] != 0
        ps = prompt_speech_token.reshape(-1).to(dev, torch.int32)
        n_spk = 1 if has_spk else 0
        Ltot = 1 + n_spk + Lt + 1 + ps.numel()
        out = torch.zeros(Ltot, c.llm_dim, device=dev)
        ops.linear(enc, self.aff_w, bias=self.aff_b, out_f32=out[1 + n_spk:1 + n_spk + Lt])
        idx = torch.full((2, Ltot), -2, device=dev, dtype=torch.int32)
        idx[0, 0] = self.sos_eos
        idx[0, 1 + n_spk + Lt] = self.task_id
        idx[1, 2 + n_spk + Lt:] = ps
        ops.embedding(self.llm_embedding, idx[0], out)
        ops.embedding(self.speech_embedding, idx[1], out)
        if has_spk:
            D = embedding.shape[1]
            e_in = embedding.to(dev, torch.float32).contiguous()
            e_n = torch.zeros(1, _round_up(D, 8), device=dev, dtype=dt)
            ops.layernorm(e_in, None, None, 1e-24 / D, rms=True, out_scale=1.0 / math.sqrt(D), out_act=e_n[:, :D])   # F.normalize
            ops.gemm(e_n, self.spk_w, 1, c.llm_dim, D, lda=e_n.stride(0), bias=self.spk_b, out_f32=out[1:2], ldo32=c.llm_dim)
        return out

    def _logits_of_last(self, T):
        """Hidden state of row T-1 (prefill: causal stack over seq[:bucket(T)], filling the K / V caches; afterwards one cached
        decode step per new row), llm_decoder on it -> st['logits'][0]."""
        c, st = self.cfg, self.st
        if self.incremental and self._cached == T - 1 and T > 1:
            self.stack.decode_step(T - 1)                      # one new row against the K / V caches
            self.n_decode_steps += 1
        else:
            Tb = min(_round_up(T, self.bucket), self.max_len)
            h = self.stack.forward(self.seq[:Tb], Tb, fill_cache=T if self.incremental else 0)      # prefill
            st["h"][0].copy_(h[T - 1])
        self._cached = T if self.incremental else 0
        ops.skinny_gemm(st["h"], self.dec_wp, 1, self.Vpad, c.llm_dim, bias=self.dec_b, out_f32=st["logits"], ldo=self.Vpad)

    def _sample(self, use_uniforms, forced_ptr=None, forced_ld=0):
        c, st = self.cfg, self.st
        p = L.SampleParams()
        p.logits, p.ldl, p.V, p.B = st["logits"].data_ptr(), self.Vpad, c.speech_token_size + 1, 1
        p.eos, p.top_k, p.top_p, p.win_size, p.tau_r = c.speech_token_size, c.top_k, c.top_p, c.win_size, c.tau_r
        p.fallback_mode, p.top_p2, p.top_k2 = 1, c.top_p + 0.15, c.top_k * c.expand_scale
        p.seed = self.seed
        p.nonce = st["nonce"].data_ptr()
        p.uniforms = st["uniforms"].data_ptr() if use_uniforms else None
        # the fork's TransformerLM.sampling_ids draws ONCE (its retry loop is commented out, llm.py:157-169): an EOS sampled at
        # 0 < i < min_len simply ends decoding; only step 0 is protected, by the -inf mask on the log-probs (:227-229).
        # min_len is therefore 0 for the sampler (no redraw), unlike the Qwen2LM classes (llm.py:806-821)
        p.max_trials = 100
        p.min_len, p.max_len = st["min_len"].data_ptr(), st["max_len"].data_ptr()
        p.forced, p.forced_ld = forced_ptr, forced_ld
        p.step, p.pos, p.n_emitted, p.finished = st["step"].data_ptr(), st["pos"].data_ptr(), st["n_emitted"].data_ptr(), st["finished"].data_ptr()
        p.out_tokens, p.out_ld = st["out_tokens"].data_ptr(), self.max_len
        p.emb_table, p.emb_dim = self.speech_embedding.data_ptr(), c.llm_dim
        p.x, p.ldx = st["x"].data_ptr(), c.llm_dim
        ops.sample_ras(p)

    @torch.no_grad()
    def _run(self, text, prompt_text, prompt_speech_token, embedding, min_ratio, max_ratio, uniforms=None, forced=None,
             collect_logp: Optional[list] = None) -> Generator[int, None, None]:
        c, st, dev = self.cfg, self.st, self.device
        x0 = self.lm_input(text, prompt_text, prompt_speech_token, embedding)
        T = x0.shape[0]
        tl = text.numel()
        min_len, max_len = int(tl * min_ratio), int(tl * max_ratio)
        if T + max_len > self.max_len:
            raise ValueError("max_len too small for this request")
        self.seq.zero_()
        self.seq[:T].copy_(x0)
        self._cached = 0
        for k in ("step", "n_emitted", "finished", "pos"):
            st[k].zero_()
        st["finished"][1:].fill_(1)
        st["min_len"].fill_(0)          # no EOS redraw in this LM (see _sample)
        st["max_len"].fill_(max_len)
        if uniforms is not None:
            st["uniforms"].copy_(uniforms.to(torch.float32))
        else:
            st["nonce"].copy_(torch.randint(0, 2 ** 62, (2,), dtype=torch.int64))
        fbuf = None
        if forced is not None:
            fbuf = torch.full((1, self.max_len), -2, device=dev, dtype=torch.int32)
            fbuf[0, :len(forced)] = torch.tensor(forced, dtype=torch.int32)
        sent = 0
        for i in range(max_len):
            self._logits_of_last(T)
            if collect_logp is not None:   # the reference masks EOS on the log-softmax output, without renormalising (:227-229)
                lp = st["logits"][0, :c.speech_token_size + 1].log_softmax(-1).cpu()
                if i == 0:
                    lp[c.speech_token_size] = -float("inf")
                collect_logp.append(lp)
            if i == 0:
                st["logits"][0, c.speech_token_size] = -1e30       # "force continue decode first token": out of the sampler's softmax
            self._sample(uniforms is not None, None if fbuf is None else fbuf.data_ptr(), self.max_len)
            fin, ne = int(st["finished"][0].item()), int(st["n_emitted"][0].item())
            if fin == 3:
                raise RuntimeError("sampling reaches max_trials 100 and still get eos when ignore_eos is True, check your input!")
            if ne > sent:
                tok = int(st["out_tokens"][0, sent].item())
                self.seq[T].copy_(st["x"][0])                      # the sampler wrote speech_embedding[tok] there
                T += 1
                sent = ne
                yield tok
            if fin != 0:
                break

    @torch.no_grad()
    def inference(self, text, text_len, prompt_text, prompt_text_len, prompt_speech_token, prompt_speech_token_len, embedding,
                  sampling: int = 25, max_token_text_ratio: float = 20, min_token_text_ratio: float = 2) -> Generator[int, None, None]:
        """Reference signature (llm.py:172-185).  Generator of python ints."""
        text_len += prompt_text_len    # mutated in place, as the reference does (:190)
        yield from self._run(text, prompt_text, prompt_speech_token, embedding, min_token_text_ratio, max_token_text_ratio)

    @torch.no_grad()
    def forced_logp(self, text, prompt_text, prompt_speech_token, embedding, forced: List[int]) -> torch.Tensor:
        """(len(forced) + 1, V + 1) log-softmax rows of a teacher-forced run (parity tests)."""
        rows: list = []
        toks = list(self._run(text, prompt_text, prompt_speech_token, embedding, 0.0, float(len(forced) + 1) / max(text.numel(), 1) + 1.0,
                              forced=forced, collect_logp=rows))
        assert toks == list(forced), (toks, forced)
        return torch.stack(rows)
