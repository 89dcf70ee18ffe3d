"""Speech-token LM on MI355X — host side of the drop-in for the reference's ``Qwen2LM`` + ``Qwen2Encoder``
(/root/reference/cosyvoice/llm/llm.py:743-874).  Same state-dict key names (HF Qwen2 names under
``llm.model.*``), same generator ``inference(...)`` yielding python ints.

Design (MI355X-first, not the reference's 4-graphs-per-layer path, llm/qwen2_5.py:97-179):
  * prefill: generic MFMA GEMMs + causal GQA flash attention over the freshly written KV cache;
  * decode: ONE hipGraph per step for the whole model: per layer
      rmsnorm(+split-K slab reduce) -> skinny QKV GEMM -> RoPE + KV append -> single-query GQA attention ->
      skinny o_proj (in-place residual) -> rmsnorm -> skinny gate/up + SwiGLU -> skinny down (split-K slabs),
    then final norm -> 6564-way head -> on-device RAS sampling, which also writes the next input embedding.
    Positions, cache lengths, EOS / min-len / max-len state all live on the device: the host only replays
    the graph and polls the token buffer (keeps the llm_job thread semantics of cli/model.py:116-128).
  * weights: 16-bit, pre-packed at load into MFMA B-fragment order for the decode path (1 KiB per wave-load).
Batching: B <= 16 sequences per step share every weight read (the reference is batch-1).
"""
import copy
import math
import os
from typing import Dict, Generator, List, Optional

import torch

from . import _lib as L
from . import ops
from .config import LlmConfig

P_ = "llm.model.model."


def _round_up(x, m):
    return (x + m - 1) // m * m


class Qwen2LM:
    DOWN_KSPLIT = int(os.environ.get("CV_DOWN_KSPLIT", "4"))  # down-projection K slices (1: in-place `x +=`, no slabs)

    def __init__(self, cfg: Optional[LlmConfig] = None, dtype: torch.dtype = torch.bfloat16, device: str = "cuda",
                 max_batch: int = 8, ctx_max: int = 1024, max_out: int = 2048, top_p: float = 0.8, top_k: int = 25,
                 win_size: int = 10, tau_r: float = 0.1):
        self.cfg = cfg or LlmConfig.full()
        self.dtype, self.device = dtype, torch.device(device)
        assert max_batch <= 32 and ctx_max % 64 == 0
        self.R = 16 if max_batch <= 16 else 32   # rows of every per-step state buffer (the skinny kernels take <= 16 rows per MFMA row group, <= 2 groups)
        self.max_batch, self.ctx_max, self.max_out = max_batch, ctx_max, max_out
        self.top_p, self.top_k, self.win_size, self.tau_r = top_p, top_k, win_size, tau_r
        # repetition fallback of the sampler: 0 = ras_sampling (random over the full distribution, utils/common.py:106-112),
        # 1 = non_random_ras_sampling (:116-123): nucleus again with (top_p2, top_k2) = (top_p + 0.15, top_k * expand_scale)
        self.fallback_mode, self.top_p2, self.top_k2 = 0, 0.0, 0
        self.speech_token_size = self.cfg.speech_token_size
        self.sos_eos, self.task_id = 0, 1
        self.fp16 = False
        self.seed = 0
        # > 0: cap the weight-streaming kernels at about this many workgroups, each walking several tile groups with the next
        # group prefetched (cv_skinny_params.max_wgs).  Measured slower than single-shot workgroups at three per CU on the
        # 104-CU share tts_batches gives the decode loop (tools/llm_kernel_bench.py), so it stays off.
        self.cu_budget = 0
        self.split_norm = os.environ.get("CV_SPLIT_NORM", "1") != "0"   # decode step: post-attention RMSNorm split over o_proj / gate-up
        self.use_graph = True
        self.use_stage_abi = os.environ.get("CV_LLM_STAGE_ABI", "1") != "0"   # decode-step graph built by cv_llm_step_graph_create
        self._loaded = False
        self._graphs: Dict[int, ops.Graph] = {}
        self._prefill_ws: Dict[tuple, dict] = {}

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    def half(self):
        return self

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd, strict: bool = False):
        ops.drop_graphs(self._graphs)   # captured decode steps hold raw pointers of the weights / state buffers replaced below
        cfg, dt, dev = self.cfg, self.dtype, self.device
        f32 = lambda k: sd[k].detach().to(device=dev, dtype=torch.float32).contiguous()
        w16 = lambda t: t.detach().to(torch.float32).to(device=dev, dtype=dt).contiguous()
        I = cfg.intermediate_size
        assert I % 16 == 0 and cfg.hidden_size % 32 == 0 and cfg.head_dim == 64
        self.layers = []
        for i in range(cfg.num_layers):
            lp = f"{P_}layers.{i}."
            wqkv = torch.cat([sd[lp + "self_attn.q_proj.weight"], sd[lp + "self_attn.k_proj.weight"], sd[lp + "self_attn.v_proj.weight"]], 0)
            bqkv = torch.cat([sd[lp + "self_attn.q_proj.bias"], sd[lp + "self_attn.k_proj.bias"], sd[lp + "self_attn.v_proj.bias"]], 0)
            g, u = sd[lp + "mlp.gate_proj.weight"].float(), sd[lp + "mlp.up_proj.weight"].float()
            # prefill GEMM wants [gate16 | up16] interleaved 16-row blocks (SwiGLU epilogue pairs adjacent MFMA tiles)
            gu = torch.stack([g.view(I // 16, 16, -1), u.view(I // 16, 16, -1)], dim=1).reshape(2 * I, -1)
            lay = dict(wqkv=w16(wqkv), bqkv=bqkv.detach().to(device=dev, dtype=torch.float32).contiguous(),
                       wo=w16(sd[lp + "self_attn.o_proj.weight"]), wgu=w16(gu), wdown=w16(sd[lp + "mlp.down_proj.weight"]),
                       g_in=f32(lp + "input_layernorm.weight"), g_post=f32(lp + "post_attention_layernorm.weight"))
            lay["p_qkv"] = ops.pack_skinny(lay["wqkv"])
            lay["p_o"] = ops.pack_skinny(lay["wo"])
            lay["p_gu"] = ops.pack_skinny(w16(torch.cat([g, u], 0)), interleave=True)
            # split-RMSNorm form of the decode step: gamma folded into the weights, 1/rms applied in the kernel's epilogue
            gam = sd[lp + "post_attention_layernorm.weight"].float().unsqueeze(0)
            lay["p_gu_g"] = ops.pack_skinny(w16(torch.cat([g * gam, u * gam], 0)), interleave=True)
            lay["p_down"] = ops.pack_skinny(lay["wdown"])
            self.layers.append(lay)
        self.g_final = f32(f"{P_}norm.weight")
        self.embed_tokens = f32(f"{P_}embed_tokens.weight")
        self.llm_embedding = f32("llm_embedding.weight")
        self.speech_embedding = f32("speech_embedding.weight")
        self.dec_b = f32("llm_decoder.bias")
        self.p_dec = ops.pack_skinny(w16(sd["llm_decoder.weight"]))
        inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, cfg.head_dim, 2, dtype=torch.float32) / cfg.head_dim))
        self.inv_freq = inv.to(dev).contiguous()
        ang = torch.arange(self.ctx_max, dtype=torch.float32)[:, None] * inv[None, :]
        self.rope_table = torch.cat([ang.cos(), ang.sin()], dim=1).to(dev).contiguous()  # [ctx_max][cos 32 | sin 32]
        self._alloc_state()
        self._loaded = True
        return self

    def new_context(self) -> "Qwen2LM":
        """A second decode context over the SAME weights: own device state, KV caches, workspaces and captured graphs, so
        two utterance batches can be decoded concurrently (two host threads, two streams).  The decode loop is a chain of
        short latency-bound kernels: two interleaved chains raise the stage's throughput 1.5-1.6x on the same CUs
        (tools/llm_dual_probe.py), which model.tts_batches uses."""
        assert self._loaded
        ctx = copy.copy(self)          # shares every weight tensor (they are never written after load_state_dict)
        ctx._graphs, ctx._prefill_ws = {}, {}
        ctx._alloc_state()
        return ctx

    def _alloc_state(self):
        cfg, dt, dev = self.cfg, self.dtype, self.device
        H, I = cfg.hidden_size, cfg.intermediate_size
        qkv_dim = cfg.q_dim + 2 * cfg.kv_dim
        z = lambda *s, dtype=torch.float32: torch.zeros(*s, device=dev, dtype=dtype)
        self.Vpad = _round_up(cfg.out_vocab, 16)
        MB = self.max_batch
        R = self.R
        self.st = dict(x=z(R, H), x2=z(R, H), xn=z(R, H, dtype=dt), xb=z(R, H, dtype=dt), ssp=z((H + 15) // 16, 32), qkv=z(R, qkv_dim), q=z(R, cfg.q_dim, dtype=dt),
                       ao=z(R, cfg.q_dim, dtype=dt), h=z(R, I, dtype=dt), slabs=z(self.DOWN_KSPLIT, 32, H),
                       logits=z(R, self.Vpad), pos=z(R, dtype=torch.int32), step=z(R, dtype=torch.int32),
                       n_emitted=z(R, dtype=torch.int32), finished=z(R, dtype=torch.int32), min_len=z(R, dtype=torch.int32),
                       max_len=z(R, dtype=torch.int32), out_tokens=z(R, self.max_out, dtype=torch.int32),
                       forced=torch.full((R, self.max_out), -1, device=dev, dtype=torch.int32),
                       uniforms=z(R, 101, 2), nonce=z(2, dtype=torch.int64))
        # per-layer caches in the fragment-tiled layout of the fused decode attention (cv_kv_retile); the prefill writes one
        # layer at a time into the row-major scratch pair (what cv_rope_append / cv_attention use) and re-tiles it
        self.kcache = [z(MB, cfg.num_kv_heads, self.ctx_max, 64, dtype=dt) for _ in range(cfg.num_layers)]
        self.vtcache = [z(MB, cfg.num_kv_heads, 64, self.ctx_max, dtype=dt) for _ in range(cfg.num_layers)]
        self._k_rm = z(MB, cfg.num_kv_heads, self.ctx_max, 64, dtype=dt)
        self._vt_rm = z(MB, cfg.num_kv_heads, 64, self.ctx_max, dtype=dt)

    # ------------------------------------------------------------------ decode step (graph-capturable)
    def _head_and_sample(self, B, use_forced, use_uniforms):
        cfg, st = self.cfg, self.st
        H = cfg.hidden_size
        ops.skinny_gemm(st["xn"], self.p_dec, B, cfg.out_vocab, H, bias=self.dec_b, out_f32=st["logits"], ldo=self.Vpad,
                        max_wgs=2 * self.cu_budget)
        ops.sample_ras(self._sample_params(B, use_forced, use_uniforms))

    def _sample_params(self, B, use_forced, use_uniforms):
        cfg, st = self.cfg, self.st
        H = cfg.hidden_size
        p = L.SampleParams()
        p.logits, p.ldl, p.V, p.B = st["logits"].data_ptr(), self.Vpad, cfg.out_vocab, B
        p.eos, p.top_k, p.top_p, p.win_size, p.tau_r = cfg.speech_token_size, self.top_k, self.top_p, self.win_size, self.tau_r
        p.seed = self.seed
        p.fallback_mode, p.top_p2, p.top_k2 = self.fallback_mode, self.top_p2, self.top_k2
        p.uniforms = st["uniforms"].data_ptr() if use_uniforms else None
        p.max_trials = 100
        p.min_len, p.max_len = st["min_len"].data_ptr(), st["max_len"].data_ptr()
        p.forced, p.forced_ld = (st["forced"].data_ptr() if use_forced else None), self.max_out
        p.step, p.pos, p.n_emitted, p.finished = (st["step"].data_ptr(), st["pos"].data_ptr(), st["n_emitted"].data_ptr(),
                                                  st["finished"].data_ptr())
        p.out_tokens, p.out_ld = st["out_tokens"].data_ptr(), self.max_out
        p.emb_table, p.emb_dim = self.speech_embedding.data_ptr(), H
        p.x, p.ldx = st["x"].data_ptr(), H
        p.nonce = st["nonce"].data_ptr()
        return p

    @staticmethod
    def _rp(B):
        """Row pitch of the split-K slabs / partial-sum planes of a step over B rows (one or two 16-row MFMA groups)."""
        return 16 if B <= 16 else 32

    def _split_qkv(self, B):
        """QKV RMSNorm as its own launch (see _decode_step): CV_SPLIT_QKV_NORM=0/1 forces it, default = more than 8 rows."""
        v = os.environ.get("CV_SPLIT_QKV_NORM")
        return self.split_norm and self.DOWN_KSPLIT >= 2 and ((v == "1") if v in ("0", "1") else B > 8)

    def _step_desc(self, B, use_forced, use_uniforms):
        """The decode step as ONE descriptor for the stage-level ABI (cv_llm_step_graph_create): every weight / state pointer of
        the step.  The layer array is kept alive on the object (the library reads it at capture time only)."""
        import ctypes as C
        cfg, st = self.cfg, self.st
        arr = (L.LlmLayer * cfg.num_layers)()
        for i, lay in enumerate(self.layers):
            a = arr[i]
            a.p_qkv, a.bqkv, a.p_o, a.p_gu, a.p_down = (lay["p_qkv"].data_ptr(), lay["bqkv"].data_ptr(), lay["p_o"].data_ptr(),
                                                        lay["p_gu_g"].data_ptr(), lay["p_down"].data_ptr())
            a.g_in, a.kcache, a.vtcache = lay["g_in"].data_ptr(), self.kcache[i].data_ptr(), self.vtcache[i].data_ptr()
        d = L.LlmStepDesc()
        d.dtype, d.B, d.num_layers, d.hidden = L.TORCH_DT[self.dtype], B, cfg.num_layers, cfg.hidden_size
        d.num_heads, d.num_kv_heads, d.inter, d.ctx_max = cfg.num_heads, cfg.num_kv_heads, cfg.intermediate_size, self.ctx_max
        d.down_ksplit, d.rms_eps = self.DOWN_KSPLIT, cfg.rms_eps
        d.split_qkv_norm = int(self._split_qkv(B))
        d.layers = C.cast(arr, C.POINTER(L.LlmLayer))
        d.x, d.x2, d.xn, d.xb = st["x"].data_ptr(), st["x2"].data_ptr(), st["xn"].data_ptr(), st["xb"].data_ptr()
        d.ssp, d.n_ssp, d.qkv, d.ao, d.h = st["ssp"].data_ptr(), st["ssp"].shape[0], st["qkv"].data_ptr(), st["ao"].data_ptr(), st["h"].data_ptr()
        d.slabs, d.logits, d.vpad = st["slabs"].data_ptr(), st["logits"].data_ptr(), self.Vpad
        d.rope_table, d.g_final, d.p_dec, d.dec_b = self.rope_table.data_ptr(), self.g_final.data_ptr(), self.p_dec.data_ptr(), self.dec_b.data_ptr()
        d.out_vocab = cfg.out_vocab
        d.sample = self._sample_params(B, use_forced, use_uniforms)
        d._keep = arr
        return d

    def _new_request_nonce(self, nonce=None):
        """Fresh Philox key material for one request, drawn from torch's global (CPU) generator: consecutive requests and
        concurrent decode contexts get different streams, ``torch.manual_seed`` reproduces them, and the captured step graph
        (which only holds the buffer's address) stays valid.  The reference samples from torch's global RNG as well
        (utils/common.py:139 torch.multinomial)."""
        if nonce is None:
            self.st["nonce"].copy_(torch.randint(0, 2 ** 62, (2,), dtype=torch.int64))
        else:
            self.st["nonce"].copy_(torch.tensor([int(nonce) * 0x9E3779B97F4A7C15 % (2 ** 62), 0], dtype=torch.int64))

    def _decode_step(self, B, use_forced=False, use_uniforms=False):
        cfg, st = self.cfg, self.st
        H, I = cfg.hidden_size, cfg.intermediate_size
        qkv_dim = cfg.q_dim + 2 * cfg.kv_dim
        KS = self.DOWN_KSPLIT
        scale = 1.0 / math.sqrt(cfg.head_dim)
        mw = self.cu_budget
        # residual stream ping-pongs between x (even layers) and x2 (odd): the fused prologue of the QKV kernel reads
        # cur (+ the previous layer's down-proj slabs) and its workgroup (0,0) writes the summed residual to nxt
        if KS == 1:
            # the residual row is updated in place by the two `out +=` projections; every norm prologue reads it directly
            x = st["x"]
            for li, lay in enumerate(self.layers):
                ops.skinny_gemm(st["xn"], lay["p_qkv"], B, qkv_dim, H, bias=lay["bqkv"], out_f32=st["qkv"], ldo=qkv_dim,
                                norm=dict(x=x, gamma=lay["g_in"], eps=cfg.rms_eps))
                ops.decode_attention(st["q"], self.kcache[li], self.vtcache[li], st["pos"], 1, st["ao"], B, cfg.num_heads,
                                     cfg.num_kv_heads, self.ctx_max, scale, qkv=st["qkv"], inv_freq=self.rope_table)
                ops.skinny_gemm(st["ao"], lay["p_o"], B, H, cfg.q_dim, mode=1, out_f32=x, ldo=H)
                ops.skinny_gemm(st["xn"], lay["p_gu"], B, 2 * I, H, mode=2, out_act=st["h"], ldoa=I,
                                norm=dict(x=x, gamma=lay["g_post"], eps=cfg.rms_eps))
                ops.skinny_gemm(st["h"], lay["p_down"], B, H, I, mode=1, out_f32=x, ldo=H)
            ops.rmsnorm_reduce(x, self.g_final, cfg.rms_eps, st["xn"], B)
            self._head_and_sample(B, use_forced, use_uniforms)
            return
        if self._split_qkv(B):
            # more than 8 rows: the QKV kernel's fused prologue (each of its 72 workgroups re-reads the fp32 residual + KS slabs of
            # every row: 286 KB at 16 rows) costs more than a launch — one workgroup per row sums the slabs into the residual and
            # leaves the normalised 16-bit row (cv_rmsnorm_reduce), the QKV kernel streams plain rows (16 rows on the decode loops'
            # 64 CUs: 10.7 -> 3.4 + 4.3 us, tools/llm_kernel_bench.py); the residual is then updated in place, no ping-pong
            x = st["x"]
            for li, lay in enumerate(self.layers):
                if li > 0:
                    ops.rmsnorm_reduce(x, lay["g_in"], cfg.rms_eps, st["xn"], B, slabs=st["slabs"], nslab=KS, slab_stride=self._rp(B) * H, ld_slab=H)
                else:
                    ops.rmsnorm_reduce(x, lay["g_in"], cfg.rms_eps, st["xn"], B)
                ops.skinny_gemm(st["xn"], lay["p_qkv"], B, qkv_dim, H, bias=lay["bqkv"], out_f32=st["qkv"], ldo=qkv_dim)
                ops.decode_attention(st["q"], self.kcache[li], self.vtcache[li], st["pos"], 1, st["ao"], B, cfg.num_heads,
                                     cfg.num_kv_heads, self.ctx_max, scale, qkv=st["qkv"], inv_freq=self.rope_table)
                ops.skinny_gemm(st["ao"], lay["p_o"], B, H, cfg.q_dim, mode=1, out_f32=x, ldo=H,
                                split_out=dict(xb=st["xb"], ss=st["ssp"]))
                ops.skinny_gemm(st["xb"], lay["p_gu_g"], B, 2 * I, H, mode=2, out_act=st["h"], ldoa=I,
                                split_in=dict(rs=st["ssp"], n=st["ssp"].shape[0], eps=cfg.rms_eps))
                ops.skinny_gemm(st["h"], lay["p_down"], B, H, I, ksplit=KS, out_f32=st["slabs"], ldo=H, slab_stride=self._rp(B) * H, max_wgs=mw)
            ops.rmsnorm_reduce(x, self.g_final, cfg.rms_eps, st["xn"], B, slabs=st["slabs"], nslab=KS, slab_stride=self._rp(B) * H, ld_slab=H)
            self._head_and_sample(B, use_forced, use_uniforms)
            return
        cur, nxt = st["x"], st["x2"]
        for li, lay in enumerate(self.layers):
            nrm = dict(x=cur, gamma=lay["g_in"], eps=cfg.rms_eps, x_out=nxt)
            if li > 0:
                nrm.update(slabs=st["slabs"], nslab=KS, slab_stride=self._rp(B) * H, ld_slab=H)
            ops.skinny_gemm(st["xn"], lay["p_qkv"], B, qkv_dim, H, bias=lay["bqkv"], out_f32=st["qkv"], ldo=qkv_dim, norm=nrm)
            ops.decode_attention(st["q"], self.kcache[li], self.vtcache[li], st["pos"], 1, st["ao"], B, cfg.num_heads,
                                 cfg.num_kv_heads, self.ctx_max, scale, qkv=st["qkv"], inv_freq=self.rope_table)
            if self.split_norm:
                # post-attention RMSNorm split over the two launches: o_proj also leaves the updated rows as 16-bit and its
                # workgroups' partial sums of squares; gate/up reads those rows straight into MFMA fragments (no prologue, no
                # LDS image, no barrier) and applies 1/rms in its epilogue (gamma is folded into the packed weights)
                ops.skinny_gemm(st["ao"], lay["p_o"], B, H, cfg.q_dim, mode=1, out_f32=nxt, ldo=H,
                                split_out=dict(xb=st["xb"], ss=st["ssp"]))
                ops.skinny_gemm(st["xb"], lay["p_gu_g"], B, 2 * I, H, mode=2, out_act=st["h"], ldoa=I,
                                split_in=dict(rs=st["ssp"], n=st["ssp"].shape[0], eps=cfg.rms_eps))
            else:
                ops.skinny_gemm(st["ao"], lay["p_o"], B, H, cfg.q_dim, mode=1, out_f32=nxt, ldo=H)
                ops.skinny_gemm(st["xn"], lay["p_gu"], B, 2 * I, H, mode=2, out_act=st["h"], ldoa=I,
                                norm=dict(x=nxt, gamma=lay["g_post"], eps=cfg.rms_eps), max_wgs=mw)
            ops.skinny_gemm(st["h"], lay["p_down"], B, H, I, ksplit=KS, out_f32=st["slabs"], ldo=H, slab_stride=self._rp(B) * H, max_wgs=mw)
            cur, nxt = nxt, cur
        ops.rmsnorm_reduce(cur, self.g_final, cfg.rms_eps, st["xn"], B, slabs=st["slabs"], nslab=KS, slab_stride=self._rp(B) * H, ld_slab=H)
        self._head_and_sample(B, use_forced, use_uniforms)

    def _step(self, B, use_forced, use_uniforms):
        if not self.use_graph:
            self._decode_step(B, use_forced, use_uniforms)
            return
        key = (B, use_forced, use_uniforms, self.seed, self.cu_budget, self.top_p, self.top_k, self.fallback_mode, self.top_p2, self.top_k2,
               self.split_norm, self.use_stage_abi)
        g = self._graphs.get(key)
        if g is None:
            if self.use_stage_abi and self.split_norm and self.DOWN_KSPLIT >= 2 and self.cu_budget == 0:
                # the whole step composed and captured inside the library (cv_llm_step_graph_create): same launches as
                # _decode_step, which stays as the eager path and as the cross-check of the C composition (tests)
                g = ops.Graph.from_llm_step(self._step_desc(B, use_forced, use_uniforms))
            else:
                g = ops.Graph().capture(lambda: self._decode_step(B, use_forced, use_uniforms))
            self._graphs[key] = g
        g.launch()

    # ------------------------------------------------------------------ prefill
    def _prefill(self, B, Lp, use_forced, use_uniforms, lens=None):
        """Input embeddings already in ws['x'] (B*Lp, H) fp32.  Writes the KV caches for positions [0,Lp), leaves the
        last position's hidden state in st['x'] and runs head + sampling (llm.py:861-866, first loop iteration).
        ``lens`` (list of B ints <= Lp): ragged batch, sequence b occupies rows [0, lens[b]) of its Lp-row slot (the rest are
        zero embeddings): attention is limited to its own keys, its decode state starts at lens[b]; the cache entries the
        padded rows write beyond lens[b] are never read (decode attends to ctx_len keys) and get overwritten as b grows."""
        cfg, dt, dev, st = self.cfg, self.dtype, self.device, self.st
        H, I = cfg.hidden_size, cfg.intermediate_size
        qkv_dim = cfg.q_dim + 2 * cfg.kv_dim
        ws = self._prefill_workspace(B, Lp)
        rows = B * Lp
        x = ws["x"]
        st["pos"].zero_()
        scale = 1.0 / math.sqrt(cfg.head_dim)
        klen = None
        if lens is not None and any(l != Lp for l in lens):
            klen = torch.tensor(list(lens), dtype=torch.int32, device=dev)
        for li, lay in enumerate(self.layers):
            ops.layernorm(x, lay["g_in"], None, cfg.rms_eps, rms=True, out_act=ws["xn"])
            ops.linear(ws["xn"], lay["wqkv"], bias=lay["bqkv"], out_f32=ws["qkv"])
            ops.rope_append(ws["qkv"], st["pos"], rows, Lp, cfg.num_heads, cfg.num_kv_heads, self.inv_freq, ws["q"], self._k_rm,
                            self._vt_rm, self.ctx_max)
            ops.attention(ws["q"], self._k_rm, self._vt_rm, ws["ao"], B=B, H=cfg.num_heads, Hkv=cfg.num_kv_heads, Tq=Lp,
                          Tk=Lp, scale=scale, q_bs=Lp * cfg.q_dim, ldq=cfg.q_dim, k_bs=cfg.num_kv_heads * self.ctx_max * 64,
                          k_hs=self.ctx_max * 64, ldk=64, vt_ld=self.ctx_max, o_bs=Lp * cfg.q_dim, ldo=cfg.q_dim, causal=True,
                          klen=klen)
            ops.kv_retile(self._k_rm, self._vt_rm, self.kcache[li], self.vtcache[li], B, cfg.num_kv_heads, self.ctx_max, Lp)
            ops.linear(ws["ao"], lay["wo"], res=x, out_f32=x)
            ops.layernorm(x, lay["g_post"], None, cfg.rms_eps, rms=True, out_act=ws["xn"])
            ops.linear(ws["xn"], lay["wgu"], act=ops.ACT_SWIGLU, out_act=ws["h"])
            ops.linear(ws["h"], lay["wdown"], res=x, out_f32=x)
        # last position of every sequence -> decode state
        if klen is None:
            st["x"][:B].copy_(x.view(B, Lp, H)[:, Lp - 1])
            st["pos"][:B].fill_(Lp - 1)  # the sampler's +1 then makes pos = Lp = cache length
        else:
            last = (klen - 1).long()
            st["x"][:B].copy_(x.view(B, Lp, H)[torch.arange(B, device=dev), last])
            st["pos"][:B].copy_(klen - 1)
        ops.rmsnorm_reduce(st["x"], self.g_final, cfg.rms_eps, st["xn"], B)
        self._head_and_sample(B, use_forced, use_uniforms)

    def _prefill_workspace(self, B, Lp):
        key = (B, Lp)
        ops.bound_cache(self._prefill_ws, key)
        if key not in self._prefill_ws:
            cfg, dt, dev = self.cfg, self.dtype, self.device
            rows = B * Lp
            e = lambda *s, dtype=torch.float32: torch.empty(*s, device=dev, dtype=dtype)
            self._prefill_ws[key] = dict(x=e(rows, cfg.hidden_size), xn=e(rows, cfg.hidden_size, dtype=dt),
                                         qkv=e(rows, cfg.q_dim + 2 * cfg.kv_dim), q=e(rows, cfg.q_dim, dtype=dt),
                                         ao=e(rows, cfg.q_dim, dtype=dt), h=e(rows, cfg.intermediate_size, dtype=dt),
                                         idx=torch.empty(3, rows, device=dev, dtype=torch.int32))
        return self._prefill_ws[key]

    def _assemble_inputs(self, ws, texts, prompt_texts, prompt_speech, B, Lp):
        """lm_input = [sos_eos, embed(prompt_text + text), task_id, speech_emb(prompt_speech)] (llm.py:837-852)."""
        idx = ws["idx"].view(3, B, Lp)
        idx.fill_(-2)
        dev = self.device
        for b in range(B):
            t = torch.cat([prompt_texts[b].reshape(-1).to(dev), texts[b].reshape(-1).to(dev)]).to(torch.int32)
            ps = prompt_speech[b].reshape(-1).to(dev, torch.int32)
            lt, lps = t.numel(), ps.numel()
            n = 1 + lt + 1 + lps
            assert n <= Lp
            idx[2, b, 0] = self.sos_eos
            idx[0, b, 1:1 + lt] = t
            idx[2, b, 1 + lt] = self.task_id
            idx[1, b, 2 + lt:n] = ps
            idx[0, b, n:] = -1   # ragged batch: rows beyond this sequence are zero embeddings
        for k, table in enumerate((self.embed_tokens, self.speech_embedding, self.llm_embedding)):
            ops.embedding(table, ws["idx"][k], ws["x"])

    def _assemble(self, ws, texts, prompt_texts, prompt_speech, B, Lp, lm_inputs):
        if lm_inputs is None:
            return self._assemble_inputs(ws, texts, prompt_texts, prompt_speech, B, Lp)
        x = ws["x"].view(B, Lp, -1)
        x.zero_()   # rows beyond a sequence's own length are zero embeddings
        for b, e in enumerate(lm_inputs):
            x[b, :e.shape[0]].copy_(e.to(self.device, torch.float32))

    # ------------------------------------------------------------------ public API
    @torch.no_grad()
    def generate_batch(self, texts: List[torch.Tensor], prompt_texts: List[torch.Tensor], prompt_speech: List[torch.Tensor],
                       forced: Optional[List[List[int]]] = None, uniforms: Optional[torch.Tensor] = None,
                       max_token_text_ratio: float = 20, min_token_text_ratio: float = 2, steps_per_poll: int = 16,
                       max_steps: Optional[int] = None, prefill_stream=None,
                       lm_inputs: Optional[List[torch.Tensor]] = None) -> List[List[int]]:
        """Run B sequences (any mix of text / prompt lengths) to completion; returns the emitted token lists.
        ``lm_inputs``: per sequence a ready-made prefill embedding sequence (L_b, hidden) fp32 instead of the
        [sos, embed(prompt_text + text), task_id, speech_emb(prompt_speech)] assembly — the hook for the fork's other LM
        front-ends, which differ from Qwen2LM only in how lm_input is built (e.g. Qwen2LM_Phoneme_Src2, llm.py:1687-1745:
        [sos, spk, fused phoneme sequence, task_id, prompt speech]); ``texts`` still gives the min/max length ratios.
        ``prefill_stream``: run the prefill (throughput-bound GEMMs / attention over B*Lp rows) on that stream instead of the
        current one — model.tts_batches hands it a stream on the flow stage's CUs, so the few-CU decode partition only ever
        runs the latency-bound token loop."""
        assert self._loaded
        B = len(texts)
        assert 1 <= B <= self.max_batch
        st = self.st
        if lm_inputs is not None:
            assert len(lm_inputs) == B and all(e.dim() == 2 and e.shape[1] == self.cfg.hidden_size for e in lm_inputs)
            lens = [int(e.shape[0]) for e in lm_inputs]
        else:
            lens = [1 + prompt_texts[b].numel() + texts[b].numel() + 1 + prompt_speech[b].numel() for b in range(B)]
        Lp = max(lens)   # ragged batches are left-aligned in Lp-row slots
        ws = self._prefill_workspace(B, Lp)
        for k in ("step", "n_emitted", "finished"):
            st[k].zero_()
        st["finished"][B:].fill_(1)
        mn = torch.zeros(self.R, dtype=torch.int32)
        mx = torch.zeros(self.R, dtype=torch.int32)
        for b in range(B):
            tl = texts[b].numel()  # text_len - prompt_text_len (llm.py:855-856)
            mn[b] = int(tl * min_token_text_ratio)
            mx[b] = int(tl * max_token_text_ratio)
            assert lens[b] + int(mx[b]) <= self.ctx_max and Lp <= self.ctx_max, "ctx_max too small"
        st["min_len"].copy_(mn)
        st["max_len"].copy_(mx)
        use_forced = forced is not None
        if use_forced:
            f = torch.full((self.R, self.max_out), -2, dtype=torch.int32)
            for b in range(B):
                f[b, :len(forced[b])] = torch.tensor(forced[b], dtype=torch.int32)
            st["forced"].copy_(f)
        use_uniforms = uniforms is not None
        if use_uniforms:
            st["uniforms"].copy_(uniforms.to(torch.float32))
        else:
            self._new_request_nonce()
        if prefill_stream is not None:
            cur = torch.cuda.current_stream()
            prefill_stream.wait_stream(cur)            # the state resets above
            with torch.cuda.stream(prefill_stream):
                self._assemble(ws, texts, prompt_texts, prompt_speech, B, Lp, lm_inputs)
                self._prefill(B, Lp, use_forced, use_uniforms, lens)
            cur.wait_stream(prefill_stream)            # KV caches, first token and decode state are in place
        else:
            self._assemble(ws, texts, prompt_texts, prompt_speech, B, Lp, lm_inputs)
            self._prefill(B, Lp, use_forced, use_uniforms, lens)
        limit = int(mx[:B].max()) if max_steps is None else max_steps
        done_steps = 1
        while done_steps < limit:
            n = min(steps_per_poll, limit - done_steps)
            for _ in range(n):
                self._step(B, use_forced, use_uniforms)
            done_steps += n
            if bool((st["finished"][:B] != 0).all().item()):
                break
        fin = st["finished"][:B].cpu()
        if bool((fin == 3).any()):
            raise RuntimeError("sampling reaches max_trials 100 and still get eos when ignore_eos is True, check your input!")
        ne = st["n_emitted"][:B].cpu()
        toks = st["out_tokens"][:B].cpu()
        return [toks[b, :int(ne[b])].tolist() for b in range(B)]

    @torch.no_grad()
    def inference(self, text, text_len, prompt_text, prompt_text_len, prompt_speech_token, prompt_speech_token_len, embedding,
                  sampling: int = 25, max_token_text_ratio: float = 20, min_token_text_ratio: float = 2,
                  lm_input: Optional[torch.Tensor] = None, nonce: Optional[int] = None) -> Generator[int, None, None]:
        """Reference signature (llm.py:823-836).  Yields python ints as they become available (polled every 8 steps).
        ``lm_input`` (L, hidden): a ready-made prefill embedding sequence (see generate_batch)."""
        assert self._loaded
        text_len += prompt_text_len  # the reference mutates text_len in place (llm.py:839)
        st = self.st
        B = 1
        Lp = int(lm_input.shape[0]) if lm_input is not None else 1 + prompt_text.numel() + text.numel() + 1 + prompt_speech_token.numel()
        ws = self._prefill_workspace(B, Lp)
        for k in ("step", "n_emitted", "finished"):
            st[k].zero_()
        st["finished"][B:].fill_(1)
        tl = text.numel()
        min_len, max_len = int(tl * min_token_text_ratio), int(tl * max_token_text_ratio)
        if Lp + max_len > self.ctx_max:
            raise ValueError("ctx_max too small for this request")
        st["min_len"].fill_(min_len)
        st["max_len"].fill_(max_len)
        self._new_request_nonce(nonce)
        self._assemble(ws, [text], [prompt_text], [prompt_speech_token], B, Lp, None if lm_input is None else [lm_input])
        self._prefill(B, Lp, False, False)
        sent, steps = 0, 1
        while True:
            fin = int(st["finished"][0].item())
            ne = int(st["n_emitted"][0].item())
            if ne > sent:
                for t in st["out_tokens"][0, sent:ne].tolist():
                    yield t
                sent = ne
            if fin == 3:
                raise RuntimeError("sampling reaches max_trials 100 and still get eos when ignore_eos is True, check your input!")
            if fin != 0 or steps >= max_len:
                break
            n = min(8, max_len - steps)
            for _ in range(n):
                self._step(B, False, False)
            steps += n

    # teacher-forced log-probs for parity tests (not on the product path of tts())
    @torch.no_grad()
    def forced_logits(self, text, prompt_text, prompt_speech, forced: List[int], rows: int = 1,
                      keep: Optional[List[int]] = None) -> torch.Tensor:
        """Returns (len(forced)+1, V) log-softmax rows: prefill + one row per forced token (SURVEY.md H1 parity contract).
        ``rows`` > 1: the same sequence in every one of ``rows`` batch rows (the launch shapes of a batched token loop: 8-row,
        16-row and two-row-group skinny kernels, batched decode attention), result (n, rows, V).  ``keep``: only these step
        indices are returned (in that order)."""
        st = self.st
        B = rows
        assert 1 <= B <= self.max_batch and len(forced) + 1 <= self.max_out
        Lp = 1 + prompt_text.numel() + text.numel() + 1 + prompt_speech.numel()
        assert Lp + len(forced) + 1 <= self.ctx_max, "ctx_max too small"
        ws = self._prefill_workspace(B, Lp)
        for k in ("step", "n_emitted", "finished"):
            st[k].zero_()
        st["finished"][B:].fill_(1)
        st["min_len"].fill_(0)
        st["max_len"].fill_(len(forced) + 8)
        f = torch.full((self.R, self.max_out), -2, dtype=torch.int32)
        f[:B, :len(forced)] = torch.tensor(forced, dtype=torch.int32)
        st["forced"].copy_(f)
        self._assemble_inputs(ws, [text] * B, [prompt_text] * B, [prompt_speech] * B, B, Lp)
        want = set(range(len(forced) + 1)) if keep is None else set(keep)
        got = {}
        self._prefill(B, Lp, True, False)
        if 0 in want:
            got[0] = st["logits"][:B, :self.cfg.out_vocab].clone()
        for i in range(1, len(forced) + 1):
            self._step(B, True, False)
            if i in want:
                got[i] = st["logits"][:B, :self.cfg.out_vocab].clone()
        order = list(range(len(forced) + 1)) if keep is None else list(keep)
        lp = torch.stack([got[i] for i in order]).log_softmax(dim=-1)
        return lp[:, 0] if rows == 1 else lp
