"""Oracle for the prefill front-end of Qwen2LM_Phoneme_Src2 (TEST INFRASTRUCTURE — see oracle/__init__.py).

Restates /root/reference/cosyvoice/llm/llm.py:1687-1745 (inference up to lm_input), :1532-1539 (encode),
transformer/encoder.py:109-172 + :388-474 (ConformerEncoder, rel-pos, no cnn, no macaron; layers shared with oracle.flow),
transformer/decoder_layer.py:60-132 (DecoderLayer, normalize_before=True), transformer/attention.py:36-135
(MultiHeadedAttention) and the fork's sampler utils/common.py:116-123 (non_random_ras_sampling)."""
import math

import torch
import torch.nn.functional as F

from . import flow as of
from . import llm as ol


def mha(sd, name, q_in, kv_in, mask, heads):
    """MultiHeadedAttention.forward (attention.py:102-135): mask (B,1,Tk) bool, True = keep."""
    B, Tq, D = q_in.shape
    dk = D // heads
    q = of._lin(sd, f"{name}.linear_q", q_in).view(B, Tq, heads, dk).transpose(1, 2)
    k = of._lin(sd, f"{name}.linear_k", kv_in).view(B, -1, heads, dk).transpose(1, 2)
    v = of._lin(sd, f"{name}.linear_v", kv_in).view(B, -1, heads, dk).transpose(1, 2)
    s = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(dk)
    m = mask.unsqueeze(1).eq(0)
    s = s.masked_fill(m, -float("inf"))
    a = torch.softmax(s, dim=-1).masked_fill(m, 0.0)
    x = torch.matmul(a, v).transpose(1, 2).contiguous().view(B, Tq, D)
    return of._lin(sd, f"{name}.linear_out", x)


def decoder_layer(sd, name, tgt, tgt_mask, memory, memory_mask, heads):
    """DecoderLayer.forward, cache=None, normalize_before=True (decoder_layer.py:91-127); FFN activation ReLU."""
    x = tgt + mha(sd, f"{name}.self_attn", *(2 * [of._ln(sd, f"{name}.norm1", tgt, 1e-5)]), tgt_mask, heads)
    x = x + mha(sd, f"{name}.src_attn", of._ln(sd, f"{name}.norm2", x, 1e-5), memory, memory_mask, heads)
    h = of._ln(sd, f"{name}.norm3", x, 1e-5)
    return x + of._lin(sd, f"{name}.feed_forward.w_2", F.relu(of._lin(sd, f"{name}.feed_forward.w_1", h)))


def conformer_encoder(sd, pcfg, xs, xs_lens, prefix="text_encoder."):
    """ConformerEncoder.forward with decoding_chunk_size=-1 (full attention), encoder.py:109-172."""
    T = xs.shape[1]
    masks = ~of.make_pad_mask(xs_lens, T).unsqueeze(1)
    xs, pos_emb = of.embed(sd, f"{prefix}embed", xs, pcfg.enc_dim)
    for i in range(pcfg.enc_blocks):
        xs = of.conformer_layer(sd, f"{prefix}encoders.{i}", xs, masks, pos_emb, pcfg.enc_heads)
    return of._ln(sd, f"{prefix}after_norm", xs, 1e-5), masks


def phoneme_lm_input(sd, pcfg, lcfg, text, pho, prompt_text, prompt_pho, prompt_speech_token, embedding):
    """llm.py:1700-1745.  text / prompt_text (1,L) BPE ids; pho / prompt_pho (1,P,4) phoneme factors; prompt_speech_token (1,N);
    embedding (0|1, D_spk).  Returns lm_input (1, 1 + [1] + P + 1 + N, H)."""
    text = torch.cat([prompt_text, text], dim=1).long()
    pho = torch.cat([prompt_pho, pho], dim=1).long()
    embs = []
    for i in range(4):
        e = F.embedding(pho[:, :, i], sd[f"text_embedding.{i}.weight"])
        if not pcfg.use_frontend_prsd and i == 3:
            e = e * 0.0
        embs.append(e)
    x = torch.cat(embs, dim=-1)
    plen = torch.tensor([x.shape[1]])
    x, _ = conformer_encoder(sd, pcfg, x, plen)
    x = of._lin(sd, "text_encoder_affine_layer", x)
    t = F.embedding(text, sd["llm.model.model.embed_tokens.weight"])
    tmask = torch.ones(1, 1, t.shape[1], dtype=torch.bool)
    pmask = torch.ones(1, 1, x.shape[1], dtype=torch.bool)
    x = decoder_layer(sd, "src_attention.0", x, pmask, t, tmask, pcfg.src_heads)
    if embedding.shape[0] != 0:
        spk = of._lin(sd, "spk_embed_affine_layer", F.normalize(embedding.float(), dim=1)).unsqueeze(1)
    else:
        spk = torch.zeros(1, 0, lcfg.hidden_size)
    sos = sd["llm_embedding.weight"][0].reshape(1, 1, -1)
    task = sd["llm_embedding.weight"][1].reshape(1, 1, -1)
    pe = F.embedding(prompt_speech_token.long(), sd["speech_embedding.weight"]) if prompt_speech_token.shape[1] else torch.zeros(1, 0, lcfg.hidden_size)
    return torch.cat([sos, spk, x, task, pe], dim=1)


def non_random_ras_sampling(weighted_scores, decoded_tokens, u_pair, top_p=0.8, top_k=10, win_size=10, tau_r=0.1, expand_scale=2):
    """utils/common.py:116-123 with injected uniforms (nucleus draw, fallback nucleus draw)."""
    top = ol.nucleus_sampling(weighted_scores, u_pair[0], top_p, top_k)
    rep = sum(1 for t in decoded_tokens[-win_size:] if t == top)
    if rep >= win_size * tau_r:
        top = ol.nucleus_sampling(weighted_scores, u_pair[1], top_p + 0.15, top_k * expand_scale)
    return top
