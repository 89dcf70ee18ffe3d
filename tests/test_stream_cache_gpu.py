"""GPU: the streaming encoder cache.  Under CosyVoice2Model's chunk mask (transformer/upsample_encoder.py:273-296,
utils/mask.py:127-200: static_chunk_size 50 tokens / 100 frames, cli/model.py:312-315) everything the encoder computes for positions
before floor((n - 3) / chunk) * chunk is final, so a streaming request re-encodes only the rows behind that prefix
(UpsampleConformerEncoder.forward_tokens_cached) against per-layer K / V^T kept from its previous chunk calls.  The reference re-encodes
everything per chunk (cli/model.py:380-407): the cached path must give the same encoder output and the same mel."""
import pytest
import torch

from cosyvoice_amd.config import FlowConfig
from cosyvoice_amd.weights import flow_state_dict

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dt,tol", [(torch.float16, 4e-3), (torch.bfloat16, 4e-2)])
@pytest.mark.parametrize("bucket", [0, 5])
def test_cached_encoder_equals_full_pass(dt, tol, bucket):
    """Encoder alone, tiny depth, chunk 10: a request growing by 10 tokens per call (+ 3 look-ahead, as CosyVoice2Model.tts schedules
    it), with and without length-bucket padding: after_norm output of ALL frames == the full pass over the same tokens."""
    from cosyvoice_amd import ops
    from cosyvoice_amd.flow import CausalMaskedDiffWithXvec
    cfg = FlowConfig.tiny()
    flow = CausalMaskedDiffWithXvec(cfg, dtype=dt).load_state_dict(flow_state_dict(cfg))
    enc = flow.encoder
    enc.static_chunk_size = 10
    g = torch.Generator().manual_seed(2)
    n_p, total = 14, 51
    toks = torch.randint(0, cfg.vocab_size, (n_p + total,), generator=g, dtype=torch.int32).cuda()
    ec = enc.new_stream_cache(256)
    D = cfg.enc_dim
    worst, encoded = 0.0, []
    lens = [n_p + 13, n_p + 23, n_p + 33, n_p + 43, n_p + total]
    for nv in lens:
        N = -(-nv // bucket) * bucket if bucket else nv
        idx = torch.full((N,), -1, dtype=torch.int32, device="cuda")
        idx[:nv] = toks[:nv]
        klen = torch.tensor([nv], dtype=torch.int32, device="cuda") if bucket else None
        ws = enc._workspace(1, N)
        ops.embedding(flow.emb_table, idx, ws["tok"].view(N, D))
        before = ec.rows_encoded
        xa_c = enc.forward_tokens_cached(ec, ws["tok"], N, nv, klen=klen).clone()
        encoded.append(ec.rows_encoded - before)
        enc.forward_tokens(ws["tok"], 1, N, klen=klen, n_valid=nv if bucket else None)
        xa_f = ws["b"]["xa"].view(2 * N, D)[: 2 * nv].float()
        d = (xa_c[: 2 * nv].float() - xa_f).abs().max().item()
        worst = max(worst, d)
        assert torch.isfinite(xa_c[: 2 * nv].float()).all()
    print(f"cached vs full encoder [{dt}, bucket {bucket}]: Linf {worst:.3e} (values ~ {xa_f.abs().mean().item():.2f}); rows encoded per call {encoded}")
    assert worst < tol
    # first call everything, afterwards only the rows behind the stable prefix floor((n_prev - 3) / 10) * 10
    exp, done = [], 0
    for nv in lens:
        N = -(-nv // bucket) * bucket if bucket else nv
        exp.append(N - done)
        done = ((nv - 3) // 10) * 10
    assert encoded == exp, (encoded, exp)


def test_streaming_mel_with_encoder_cache_equals_recompute(monkeypatch):
    """Full-depth flow, the C4 prompt (250 tokens) and the streaming schedule of cli/model.py:380-407 (hop 50 + 3 look-ahead, then the
    tail): every chunk's mel with the encoder cache == without it, and the cache re-encodes 53-ish rows per chunk instead of 300+."""
    from cosyvoice_amd.flow import CausalMaskedDiffWithXvec
    cfg = FlowConfig.full()
    flow = CausalMaskedDiffWithXvec(cfg, dtype=torch.float16).load_state_dict(flow_state_dict(cfg))
    flow.encoder.static_chunk_size = 50
    flow.decoder.use_graph = True
    flow.length_bucket = 25
    g = torch.Generator().manual_seed(9)
    n_p, n_g = 250, 167
    ptok = torch.randint(0, cfg.vocab_size, (1, n_p), generator=g, dtype=torch.int32)
    tok = torch.randint(0, cfg.vocab_size, (1, n_g), generator=g, dtype=torch.int32)
    pfeat = torch.clamp(torch.randn(1, 2 * n_p, 80, generator=g) * 2 - 6, -11.5, 2.0)
    emb = torch.randn(1, cfg.spk_embed_dim, generator=g)
    L = lambda n: torch.tensor([n], dtype=torch.int32)
    calls = [(53, False), (103, False), (153, False), (n_g, True)]

    def run(key):
        outs = []
        for n, fin in calls:
            mel, _ = flow.inference(token=tok[:, :n], token_len=L(n), prompt_token=ptok, prompt_token_len=L(n_p), prompt_feat=pfeat,
                                    prompt_feat_len=L(2 * n_p), embedding=emb, finalize=fin, **({"cache_key": key} if key else {}))
            outs.append(mel.cpu())
        return outs
    ref = run(None)
    got = run("req-1")
    ec = flow._stream_caches["req-1"]
    worst = max((a - b).abs().max().item() for a, b in zip(got, ref))
    l1 = max((a - b).abs().mean().item() for a, b in zip(got, ref))
    print(f"streaming mel, encoder cache vs recompute: Linf {worst:.3e} L1 {l1:.3e}; token rows encoded {ec.rows_encoded} in {ec.calls} calls "
          f"(recompute: {sum(n_p + n for n, _ in calls)})")
    assert worst < 5e-3 and l1 < 2e-4
    # call 1: bucketed length 325 -> all rows; then the rows behind the stable prefix, up to the bucketed length: 375 - 300, 425 - 350, 425 - 400
    assert ec.calls == 4 and ec.rows_encoded == 325 + 75 + 75 + 25
    flow.drop_stream_cache("req-1")
    assert "req-1" not in flow._stream_caches
    monkeypatch.setenv("CV_STREAM_ENC_CACHE", "0")
    off = run("req-2")
    assert "req-2" not in flow._stream_caches and all(torch.equal(a, b) for a, b in zip(off, ref))
