"""Oracle for the Qwen2LM speech-token decoder (TEST INFRASTRUCTURE — see oracle/__init__.py).

Restates (a) the HF Qwen2 decoder stack the reference calls through
Qwen2Encoder.forward_one_step (/root/reference/cosyvoice/llm/llm.py:754-766; third-party
arithmetic: transformers, unpinned by the reference) and (b) Qwen2LM.inference
(llm.py:823-874) with the samplers of utils/common.py:109-146.
"""
import math
from typing import Callable, List, Optional

import torch
import torch.nn.functional as F

from cosyvoice_amd.config import LlmConfig

P = "llm.model.model."


def rms_norm(x, w, eps):
    v = x.float().pow(2).mean(-1, keepdim=True)
    return w * (x.float() * torch.rsqrt(v + eps))


def rope_cos_sin(cfg: LlmConfig, positions: torch.Tensor):
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, cfg.head_dim, 2, dtype=torch.float32) / cfg.head_dim))
    fr = positions.float()[:, None] * inv[None, :]
    emb = torch.cat([fr, fr], dim=-1)
    return emb.cos(), emb.sin()


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], dim=-1)


class KVCache:
    def __init__(self, n_layers):
        self.k: List[Optional[torch.Tensor]] = [None] * n_layers
        self.v: List[Optional[torch.Tensor]] = [None] * n_layers

    @property
    def length(self):
        return 0 if self.k[0] is None else self.k[0].shape[2]


def qwen2_forward(sd, cfg: LlmConfig, x: torch.Tensor, cache: KVCache) -> torch.Tensor:
    """x (B,L,H) input embeddings appended after the cache; causal attention; returns final-norm
    hidden states (B,L,H) — what Qwen2Encoder.forward_one_step returns as hidden_states[-1]."""
    B, L, H = x.shape
    past = cache.length
    pos = torch.arange(past, past + L)
    cos, sin = rope_cos_sin(cfg, pos)
    nh, nkv, dh = cfg.num_heads, cfg.num_kv_heads, cfg.head_dim
    causal = torch.ones(L, past + L, dtype=torch.bool).tril(diagonal=past)
    for i in range(cfg.num_layers):
        lp = f"{P}layers.{i}."
        h = rms_norm(x, sd[f"{lp}input_layernorm.weight"], cfg.rms_eps)
        q = F.linear(h, sd[f"{lp}self_attn.q_proj.weight"], sd[f"{lp}self_attn.q_proj.bias"]).view(B, L, nh, dh).transpose(1, 2)
        k = F.linear(h, sd[f"{lp}self_attn.k_proj.weight"], sd[f"{lp}self_attn.k_proj.bias"]).view(B, L, nkv, dh).transpose(1, 2)
        v = F.linear(h, sd[f"{lp}self_attn.v_proj.weight"], sd[f"{lp}self_attn.v_proj.bias"]).view(B, L, nkv, dh).transpose(1, 2)
        q = q * cos + rotate_half(q) * sin
        k = k * cos + rotate_half(k) * sin
        if cache.k[i] is not None:
            k = torch.cat([cache.k[i], k], dim=2)
            v = torch.cat([cache.v[i], v], dim=2)
        cache.k[i], cache.v[i] = k, v
        rep = nh // nkv
        kk = k.repeat_interleave(rep, dim=1)
        vv = v.repeat_interleave(rep, dim=1)
        s = torch.matmul(q, kk.transpose(-2, -1)) / math.sqrt(dh)
        s = s.masked_fill(~causal, float("-inf"))
        o = torch.matmul(torch.softmax(s, dim=-1), vv).transpose(1, 2).reshape(B, L, nh * dh)
        x = x + F.linear(o, sd[f"{lp}self_attn.o_proj.weight"])
        h = rms_norm(x, sd[f"{lp}post_attention_layernorm.weight"], cfg.rms_eps)
        g = F.linear(h, sd[f"{lp}mlp.gate_proj.weight"])
        u = F.linear(h, sd[f"{lp}mlp.up_proj.weight"])
        x = x + F.linear(F.silu(g) * u, sd[f"{lp}mlp.down_proj.weight"])
    return rms_norm(x, sd[f"{P}norm.weight"], cfg.rms_eps)


def build_lm_input(sd, cfg: LlmConfig, text, prompt_text, prompt_speech_token):
    """llm.py:837-852: [sos_eos, embed(prompt_text+text), task_id, speech_emb(prompt_speech_token)]."""
    text = torch.cat([prompt_text, text], dim=1).long()
    te = F.embedding(text, sd[f"{P}embed_tokens.weight"])
    sos = sd["llm_embedding.weight"][0].reshape(1, 1, -1)
    task = sd["llm_embedding.weight"][1].reshape(1, 1, -1)
    if prompt_speech_token.shape[1] != 0:
        pe = F.embedding(prompt_speech_token.long(), sd["speech_embedding.weight"])
    else:
        pe = torch.zeros(1, 0, cfg.hidden_size)
    return torch.cat([sos, te, task, pe], dim=1)


def logits_to_logp(sd, y_last):
    return F.linear(y_last, sd["llm_decoder.weight"], sd["llm_decoder.bias"]).log_softmax(dim=-1)


# ------------------------------------------------------------------ samplers (utils/common.py:109-146)
def _inverse_cdf(prob: torch.Tensor, u: float) -> int:
    """Draw from a categorical given a uniform u in [0,1) (stands in for torch.multinomial, whose RNG
    stream cannot be reproduced across implementations — SURVEY.md H1)."""
    c = torch.cumsum(prob.double() / prob.double().sum(), 0)
    idx = int(torch.searchsorted(c, torch.tensor(u, dtype=torch.float64), right=True).item())
    return min(idx, prob.numel() - 1)


def nucleus_candidates(weighted_scores: torch.Tensor, top_p=0.8, top_k=25):
    # utils/common.py:126-141: stable descending sort; take while cum<top_p and n<top_k
    sorted_value, sorted_idx = weighted_scores.softmax(dim=0).sort(descending=True, stable=True)
    prob, indices, cum = [], [], 0.0
    for i in range(len(sorted_idx)):
        if cum < top_p and len(prob) < top_k:
            cum += sorted_value[i]
            prob.append(sorted_value[i])
            indices.append(sorted_idx[i])
        else:
            break
    return torch.tensor(prob), torch.tensor(indices, dtype=torch.long)


def nucleus_sampling(weighted_scores, u: float, top_p=0.8, top_k=25) -> int:
    prob, idx = nucleus_candidates(weighted_scores, top_p, top_k)
    return int(idx[_inverse_cdf(prob, u)])


def random_sampling(weighted_scores, u: float) -> int:
    return _inverse_cdf(weighted_scores.softmax(dim=0), u)


def ras_sampling(weighted_scores, decoded_tokens, u_pair, top_p=0.8, top_k=25, win_size=10, tau_r=0.1) -> int:
    """utils/common.py:109-114.  u_pair = (u for the nucleus draw, u for the fallback draw)."""
    top = nucleus_sampling(weighted_scores, u_pair[0], top_p, top_k)
    rep = sum(1 for t in decoded_tokens[-win_size:] if t == top)
    if rep >= win_size * tau_r:
        top = random_sampling(weighted_scores, u_pair[1])
    return top


def sampling_ids(weighted_scores, decoded_tokens, ignore_eos: bool, eos: int, uniforms: Callable[[int], tuple],
                 max_trials: int = 100) -> int:
    # llm.py:806-821.  ``uniforms(trial)`` supplies the (nucleus, fallback) uniforms of redraw number ``trial``.
    num_trials = 0
    while True:
        top = ras_sampling(weighted_scores, decoded_tokens, uniforms(num_trials))
        if (not ignore_eos) or top != eos:
            return top
        num_trials += 1
        if num_trials > max_trials:
            raise RuntimeError("sampling reaches max_trials {} and still get eos when ignore_eos is True".format(max_trials))


def lm_inference(sd, cfg: LlmConfig, text, prompt_text, prompt_speech_token, uniforms: Callable[[int], tuple],
                 max_token_text_ratio=20, min_token_text_ratio=2, forced_tokens: Optional[List[int]] = None,
                 collect_logp: Optional[list] = None):
    """Qwen2LM.inference, llm.py:823-874.  Yields python ints.  ``forced_tokens`` teacher-forces the
    emitted ids (the sampler still runs) so random-weight runs have a fixed length."""
    lm_input = build_lm_input(sd, cfg, text, prompt_text, prompt_speech_token)
    text_len = text.shape[1]
    min_len = int(text_len * min_token_text_ratio)
    max_len = int(text_len * max_token_text_ratio)
    out_tokens: List[int] = []
    cache = KVCache(cfg.num_layers)
    eos = cfg.speech_token_size
    for i in range(max_len):
        y = qwen2_forward(sd, cfg, lm_input, cache)
        logp = logits_to_logp(sd, y[:, -1]).squeeze(0)
        if collect_logp is not None:
            collect_logp.append(logp.clone())
        top = sampling_ids(logp, out_tokens, ignore_eos=i < min_len, eos=eos, uniforms=uniforms)
        if forced_tokens is not None:
            if len(out_tokens) >= len(forced_tokens):
                break
            top = forced_tokens[len(out_tokens)]
        if top == eos:
            break
        if top > eos:
            continue
        yield top
        out_tokens.append(top)
        lm_input = sd["speech_embedding.weight"][top].reshape(1, 1, -1)
