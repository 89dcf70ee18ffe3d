"""Oracle for the CosyVoice-v1 TransformerLM (TEST INFRASTRUCTURE — see oracle/__init__.py).

Restates /root/reference/cosyvoice/llm/llm.py:89-97 (encode), :171-237 (inference), transformer/encoder.py:109-172
(ConformerEncoder.forward, decoding_chunk_size=1 = causal), :174-273 (forward_chunk; with an attention cache it returns, per
step, the last rows of the full causal forward pass — restated here as that full pass), transformer/encoder_layer.py:24-107
(TransformerEncoderLayer), subsampling.py:338-372 (LegacyLinearNoSubsampling).  Rel-pos attention / tables: oracle.flow."""
import math

import torch
import torch.nn.functional as F

from . import flow as of


def transformer_layer(sd, name, x, mask, pos_emb, heads):
    r = x
    x = r + of.rel_attention(sd, f"{name}.self_attn", of._ln(sd, f"{name}.norm1", x, 1e-12), mask, pos_emb, heads)
    r = x
    h = of._ln(sd, f"{name}.norm2", x, 1e-12)
    return r + of._lin(sd, f"{name}.feed_forward.w_2", F.relu(of._lin(sd, f"{name}.feed_forward.w_1", h)))


def text_encoder(sd, cfg, xs):
    """ConformerEncoder with static/decoding chunk size 1: every position attends to itself and the past."""
    T = xs.shape[1]
    xs, pos_emb = of.embed(sd, "text_encoder.embed", xs, cfg.enc_dim)
    causal = torch.ones(T, T, dtype=torch.bool).tril().unsqueeze(0)
    for i in range(cfg.enc_blocks):
        xs = of.conformer_layer(sd, f"text_encoder.encoders.{i}", xs, causal, pos_emb, cfg.enc_heads)
    return of._ln(sd, "text_encoder.after_norm", xs, 1e-5)


def lm_input(sd, cfg, text, prompt_text, prompt_speech_token, embedding):
    """llm.py:188-213 -> (1, 1 + [1] + L + 1 + N, llm_dim)."""
    t = F.embedding(torch.cat([prompt_text, text], dim=1).long(), sd["text_embedding.weight"])
    t = of._lin(sd, "text_encoder_affine_layer", text_encoder(sd, cfg, t))
    if embedding.shape[0] != 0:
        spk = of._lin(sd, "spk_embed_affine_layer", F.normalize(embedding.float(), dim=1)).unsqueeze(1)
    else:
        spk = torch.zeros(1, 0, cfg.llm_dim)
    sos = sd["llm_embedding.weight"][0].reshape(1, 1, -1)
    task = sd["llm_embedding.weight"][1].reshape(1, 1, -1)
    pe = F.embedding(prompt_speech_token.long(), sd["speech_embedding.weight"]) if prompt_speech_token.shape[1] else torch.zeros(1, 0, cfg.llm_dim)
    return torch.cat([sos, spk, t, task, pe], dim=1)


def llm_hidden(sd, cfg, seq):
    """TransformerEncoder over the whole sequence with a causal mask = what forward_chunk + att_cache computes incrementally."""
    T = seq.shape[1]
    x = F.relu(of._ln(sd, "llm.embed.out.1", of._lin(sd, "llm.embed.out.0", seq), 1e-5)) * math.sqrt(cfg.llm_dim)
    pos_emb = of.rel_pos_table(cfg.llm_dim, T)
    causal = torch.ones(T, T, dtype=torch.bool).tril().unsqueeze(0)
    for i in range(cfg.llm_blocks):
        x = transformer_layer(sd, f"llm.encoders.{i}", x, causal, pos_emb, cfg.llm_heads)
    return of._ln(sd, "llm.after_norm", x, 1e-5)


def forced_logp(sd, cfg, text, prompt_text, prompt_speech_token, embedding, forced):
    """log-softmax rows the decode loop sees when the emitted ids are forced: row i = step i (llm.py:221-236; the EOS column of
    row 0 is -inf, :227-229)."""
    seq = lm_input(sd, cfg, text, prompt_text, prompt_speech_token, embedding)
    L0 = seq.shape[1]
    if len(forced):
        seq = torch.cat([seq, F.embedding(torch.tensor(forced).long()[None], sd["speech_embedding.weight"])], dim=1)
    h = llm_hidden(sd, cfg, seq)[0, L0 - 1:]
    logp = of._lin(sd, "llm_decoder", h).log_softmax(dim=-1)
    logp[0, cfg.speech_token_size] = -float("inf")
    return logp
