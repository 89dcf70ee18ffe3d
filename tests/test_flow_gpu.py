"""GPU: the HIP flow path (cosyvoice_amd.flow, through the C ABI) against the reference-minted goldens and the
CPU oracle.  Tolerances: bf16/fp16 MFMA operands with fp32 accumulation vs an fp32 reference; the reference's own
precedent for an estimator-backend swap is rtol 1e-2 (bin/export_onnx.py:100)."""
import os

import numpy as np
import pytest
import torch

from cosyvoice_amd.config import FlowConfig
from cosyvoice_amd.weights import flow_state_dict

pytestmark = pytest.mark.gpu

DTS = [(torch.bfloat16, 4e-2, 8e-3), (torch.float16, 6e-3, 1.2e-3)]  # (dtype, L-inf tol, mean-abs tol)


def _golden(golden_dir, name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, name + ".npz")).items()}


def _report(name, got, ref):
    d = (got.float().cpu() - ref).abs()
    print(f"{name}: Linf {d.max().item():.3e} L1 {d.mean().item():.3e} ref-absmean {ref.abs().mean().item():.3e}")
    return d.max().item(), d.mean().item()


@pytest.mark.parametrize("dt,linf,l1", DTS)
def test_estimator_slot_vs_golden(golden_dir, dt, linf, l1):
    from cosyvoice_amd.flow import CausalMaskedDiffWithXvec
    for name, cfg in (("flow_tiny", FlowConfig.tiny()),
                      ("flow_est_1block", FlowConfig(est_n_blocks=1, est_mid_blocks=1, enc_blocks=1, enc_up_blocks=1, vocab_size=64))):
        g = _golden(golden_dir, name)
        flow = CausalMaskedDiffWithXvec(cfg, dtype=dt).load_state_dict(flow_state_dict(cfg))
        T = g["est_x"].shape[-1]
        out = flow.decoder.estimator(g["est_x"].cuda(), torch.ones(2, 1, T).cuda(), g["est_mu"].cuda(), g["est_t"].cuda(),
                                     g["est_spks"].cuda(), g["est_cond"].cuda())
        a, b = _report(f"estimator[{name},{dt}]", out, g["est_out"])
        assert a < linf and b < l1


@pytest.mark.parametrize("dt,linf,l1", DTS)
@pytest.mark.parametrize("chunk,key", [(0, "enc_full"), (4, "enc_chunk4")])
def test_encoder_vs_golden(golden_dir, dt, linf, l1, chunk, key):
    from cosyvoice_amd import ops
    from cosyvoice_amd.flow import CausalMaskedDiffWithXvec
    cfg = FlowConfig.tiny()
    g = _golden(golden_dir, "flow_tiny")
    flow = CausalMaskedDiffWithXvec(cfg, dtype=dt).load_state_dict(flow_state_dict(cfg))
    enc = flow.encoder
    enc.static_chunk_size = chunk
    x = g["enc_in"].cuda()
    R, N, D = x.shape
    ws = enc._workspace(R, N)
    ws["tok"].copy_(x.to(dt))
    enc.forward_tokens(ws["tok"], R, N)
    torch.cuda.synchronize()
    a, b = _report(f"encoder[{key},{dt}]", ws["b"]["xa"], g[key])
    assert a < linf * 2 and b < l1 * 2  # LayerNorm output, unit scale, stored in 16-bit


@pytest.mark.parametrize("dt,linf,l1", DTS)
@pytest.mark.parametrize("chunk,key", [(0, "mel_full"), (4, "mel_chunk4")])
def test_inference_vs_golden(golden_dir, dt, linf, l1, chunk, key):
    from cosyvoice_amd.flow import CausalMaskedDiffWithXvec
    cfg = FlowConfig.tiny()
    g = _golden(golden_dir, "flow_tiny")
    flow = CausalMaskedDiffWithXvec(cfg, dtype=dt).load_state_dict(flow_state_dict(cfg))
    flow.encoder.static_chunk_size = chunk
    n_g, n_p = g["token"].shape[1], g["prompt_token"].shape[1]
    mel, _ = flow.inference(token=g["token"], token_len=torch.tensor([n_g]), prompt_token=g["prompt_token"],
                            prompt_token_len=torch.tensor([n_p]), prompt_feat=g["prompt_feat"],
                            prompt_feat_len=torch.tensor([2 * n_p]), embedding=g["embedding"])
    assert mel.shape == g[key].shape and mel.dtype == torch.float32
    a, b = _report(f"mel[{key},{dt}]", mel, g[key])
    assert a < linf * 2 and b < l1 * 2


@pytest.mark.parametrize("dt,linf,l1", DTS)
def test_inference_vs_oracle_larger_and_batched(dt, linf, l1):
    """N_p=20, N_g=45 (T=130: two attention query tiles, ragged tile edges) against the oracle; batch of 3 equals
    three batch-1 runs; hipGraph replay equals the eager launch sequence."""
    from cosyvoice_amd.flow import CausalMaskedDiffWithXvec
    from oracle import flow as of
    cfg = FlowConfig.tiny()
    sd = flow_state_dict(cfg)
    flow = CausalMaskedDiffWithXvec(cfg, dtype=dt).load_state_dict(sd)
    g = torch.Generator().manual_seed(7)
    B, n_p, n_g = 3, 20, 45
    tok = torch.randint(0, cfg.vocab_size, (B, n_g), generator=g, dtype=torch.int32)
    ptok = torch.randint(0, cfg.vocab_size, (B, n_p), generator=g, dtype=torch.int32)
    pfeat = torch.clamp(torch.randn(B, 2 * n_p, 80, generator=g) * 2 - 6, -11.5, 2.0)
    emb = torch.randn(B, cfg.spk_embed_dim, generator=g)
    ref0 = of.inference(sd, cfg, tok[:1], ptok[:1], pfeat[:1], emb[:1])
    mb = flow.inference_batch(tok, ptok, pfeat, emb).clone()
    a, b = _report(f"mel[T=130,{dt}]", mb[:1], ref0)
    assert a < linf * 2 and b < l1 * 2
    for i in range(B):
        m1 = flow.inference_batch(tok[i:i + 1], ptok[i:i + 1], pfeat[i:i + 1], emb[i:i + 1])
        assert (m1[0] - mb[i]).abs().max().item() < 1e-5
    flow.decoder.use_graph = True
    flow.inference_batch(tok, ptok, pfeat, emb)          # captures
    mg = flow.inference_batch(tok, ptok, pfeat, emb).clone()  # replays
    torch.cuda.synchronize()
    assert (mg - mb).abs().max().item() < 1e-6


def test_inference_without_prompt_and_minimal_length():
    """Edge cases of flow.inference (flow.py:258-319): no prompt at all (N_p = 0: cond is all zeros, nothing is sliced off)
    and the shortest inputs (one generated token = two mel frames, with and without a prompt)."""
    from cosyvoice_amd.flow import CausalMaskedDiffWithXvec
    from oracle import flow as of
    cfg = FlowConfig.tiny()
    sd = flow_state_dict(cfg)
    flow = CausalMaskedDiffWithXvec(cfg, dtype=torch.float16).load_state_dict(sd)
    g = torch.Generator().manual_seed(11)
    emb = torch.randn(1, cfg.spk_embed_dim, generator=g)
    for n_p, n_g in ((0, 9), (0, 1), (3, 1)):
        tok = torch.randint(0, cfg.vocab_size, (1, n_g), generator=g, dtype=torch.int32)
        ptok = torch.randint(0, cfg.vocab_size, (1, n_p), generator=g, dtype=torch.int32)
        pfeat = torch.clamp(torch.randn(1, 2 * n_p, 80, generator=g) * 2 - 6, -11.5, 2.0)
        ref = of.inference(sd, cfg, tok, ptok, pfeat, emb)
        mel, cache = flow.inference(token=tok, token_len=torch.tensor([n_g]), prompt_token=ptok, prompt_token_len=torch.tensor([n_p]),
                                    prompt_feat=pfeat, prompt_feat_len=torch.tensor([2 * n_p]), embedding=emb, finalize=True)
        assert cache is None and mel.shape == (1, 80, 2 * n_g) and mel.dtype == torch.float32
        a, b = _report(f"mel[N_p={n_p},N_g={n_g}]", mel, ref)
        assert a < 2e-2 and b < 5e-3


def test_length_bucketing_is_exact():
    """length_bucket pads a request to the next multiple of 25 tokens with the tail masked (attention klen, zeroed look-ahead
    input): same mel as the unpadded run, and requests of different lengths inside one bucket share ONE captured graph."""
    from cosyvoice_amd.flow import CausalMaskedDiffWithXvec
    cfg = FlowConfig.tiny()
    sd = flow_state_dict(cfg)
    ref_flow = CausalMaskedDiffWithXvec(cfg, dtype=torch.float16).load_state_dict(sd)
    flow = CausalMaskedDiffWithXvec(cfg, dtype=torch.float16).load_state_dict(sd)
    flow.length_bucket = 25
    flow.decoder.use_graph = True
    g = torch.Generator().manual_seed(13)
    emb = torch.randn(1, cfg.spk_embed_dim, generator=g)
    n_p = 10
    ptok = torch.randint(0, cfg.vocab_size, (1, n_p), generator=g, dtype=torch.int32)
    pfeat = torch.clamp(torch.randn(1, 2 * n_p, 80, generator=g) * 2 - 6, -11.5, 2.0)
    for chunk in (0, 25):
        ref_flow.encoder.static_chunk_size = flow.encoder.static_chunk_size = chunk
        for n_g in (29, 33, 40, 15):                     # 39, 43, 50 tokens -> one 50-token bucket; 25 -> exact fit
            tok = torch.randint(0, cfg.vocab_size, (1, n_g), generator=g, dtype=torch.int32)
            want = ref_flow.inference_batch(tok, ptok, pfeat, emb).clone()
            got = flow.inference_batch(tok, ptok, pfeat, emb).clone()      # first call of a bucket captures, later ones replay
            got2 = flow.inference_batch(tok, ptok, pfeat, emb).clone()
            assert got.shape == want.shape == (1, 80, 2 * n_g)
            assert (got - want).abs().max().item() < 2e-4, (chunk, n_g, (got - want).abs().max().item())
            assert (got2 - want).abs().max().item() < 2e-4
    assert len(flow.decoder._graphs) == 2, len(flow.decoder._graphs)      # buckets of 25 and 50 tokens


def test_ragged_batch_equals_single_utterances():
    """inference_ragged: utterances of different prompt and token lengths in ONE batched pass (rows padded, tails masked) give
    the mel each utterance gives alone."""
    from cosyvoice_amd.flow import CausalMaskedDiffWithXvec
    cfg = FlowConfig.tiny()
    flow = CausalMaskedDiffWithXvec(cfg, dtype=torch.float16).load_state_dict(flow_state_dict(cfg))
    g = torch.Generator().manual_seed(23)
    shapes = [(10, 29), (0, 7), (4, 41), (13, 1)]          # (prompt tokens, generated tokens)
    toks = [torch.randint(0, cfg.vocab_size, (n,), generator=g, dtype=torch.int32) for _, n in shapes]
    ptoks = [torch.randint(0, cfg.vocab_size, (p,), generator=g, dtype=torch.int32) for p, _ in shapes]
    pfeats = [torch.clamp(torch.randn(2 * p, 80, generator=g) * 2 - 6, -11.5, 2.0) for p, _ in shapes]
    embs = torch.randn(len(shapes), cfg.spk_embed_dim, generator=g)
    for chunk in (0, 25):
        flow.encoder.static_chunk_size = chunk
        alone = [flow.inference_batch(toks[b][None], ptoks[b][None], pfeats[b][None], embs[b:b + 1])[0].clone() for b in range(len(shapes))]
        together = flow.inference_ragged(toks, ptoks, pfeats, embs)
        for b, (p, n) in enumerate(shapes):
            assert together[b].shape == alone[b].shape == (80, 2 * n)
            assert (together[b] - alone[b]).abs().max().item() < 2e-4, (chunk, b, (together[b] - alone[b]).abs().max().item())


@pytest.mark.gpu
def test_shape_caches_are_bounded():
    """A service sees an open-ended set of request lengths: the per-shape caches (buffers, encoder / estimator workspaces,
    position tables, captured Euler loops) are flushed together once ``shape_cache_cap`` shapes are held, and results after a
    flush equal those of a fresh model."""
    from cosyvoice_amd.config import FlowConfig
    from cosyvoice_amd.flow import CausalMaskedDiffWithXvec
    from cosyvoice_amd.weights import flow_state_dict
    fc = FlowConfig.tiny()
    sd = flow_state_dict(fc)
    m = CausalMaskedDiffWithXvec(fc, dtype=torch.float16).load_state_dict(sd)
    m.decoder.use_graph = True
    m.shape_cache_cap = 2
    g = torch.Generator().manual_seed(0)
    ptok = torch.randint(0, fc.vocab_size, (1, 6), generator=g, dtype=torch.int32)
    pfeat = torch.clamp(torch.randn(1, 12, 80, generator=g) * 2 - 6, -11.5, 2.0)
    emb = torch.randn(1, fc.spk_embed_dim, generator=g)
    toks = {n: torch.randint(0, fc.vocab_size, (1, n), generator=g, dtype=torch.int32) for n in (5, 7, 9, 11, 5, 7)}
    outs = {}
    for n, tok in toks.items():
        for rep in range(2):      # second call of a shape replays the captured loop
            mel = m.inference_batch(tok, ptok, pfeat, emb).clone()
        outs[n] = mel
        assert len(m._bufs) <= 2 and len(m.encoder._ws) <= 2 and len(m.decoder.estimator._ws) <= 2 and len(m.decoder._graphs) <= 2
    fresh = CausalMaskedDiffWithXvec(fc, dtype=torch.float16).load_state_dict(sd)
    for n, tok in toks.items():
        assert torch.equal(fresh.inference_batch(tok, ptok, pfeat, emb), outs[n])
