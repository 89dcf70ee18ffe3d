"""GPU: FULL-depth models (the BASELINE architecture: 24-layer Qwen2-0.5B shape, 56-block estimator + 10-layer encoder)
against the CPU oracle at lengths the oracle finishes in seconds, plus size-independent properties at the BASELINE
lengths (determinism, batch invariance, graph replay == eager).  Tolerances are stated per test."""
import pytest
import torch

from cosyvoice_amd.config import FlowConfig, LlmConfig
from cosyvoice_amd.weights import flow_state_dict, llm_state_dict

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full_flow():
    cfg = FlowConfig.full()
    return cfg, flow_state_dict(cfg)


@pytest.mark.parametrize("dt,l1_tol,linf_tol", [(torch.float16, 1e-3, 1.5e-2), (torch.bfloat16, 1.2e-2, 1.5e-1)])
def test_full_depth_flow_vs_oracle(full_flow, dt, l1_tol, linf_tol):
    """north_star asks for mel L1 <= 1e-3 vs the fp32 reference: asserted for fp16 operands (what the bench uses; measured
    8.1e-4) on the full 56-block estimator x 10 Euler steps with random (kaiming-scale) weights.  bf16 operands do NOT meet the
    target (measured 5.7e-3, asserted at that level): BASELINE C2's "bf16" is reported in the bench as fp16 for that reason."""
    from cosyvoice_amd.flow import CausalMaskedDiffWithXvec
    from oracle import flow as of
    cfg, sd = full_flow
    g = torch.Generator().manual_seed(5)
    n_p, n_g = 20, 30
    tok = torch.randint(0, cfg.vocab_size, (1, n_g), generator=g, dtype=torch.int32)
    ptok = torch.randint(0, cfg.vocab_size, (1, n_p), generator=g, dtype=torch.int32)
    pfeat = torch.clamp(torch.randn(1, 2 * n_p, 80, generator=g) * 2 - 6, -11.5, 2.0)
    emb = torch.randn(1, cfg.spk_embed_dim, generator=g)
    ref = of.inference(sd, cfg, tok, ptok, pfeat, emb, static_chunk_size=0)
    flow = CausalMaskedDiffWithXvec(cfg, dtype=dt).load_state_dict(sd)
    mel = flow.inference_batch(tok, ptok, pfeat, emb).cpu()
    d = (mel - ref).abs()
    print(f"full-depth flow [{dt}]: mel L1 {d.mean().item():.3e}  Linf {d.max().item():.3e}  (ref abs-mean {ref.abs().mean().item():.3f})")
    assert d.mean().item() < l1_tol and d.max().item() < linf_tol


def test_full_size_flow_properties_at_baseline_length(full_flow):
    """T = 1000 frames (10 s prompt + 10 s), batch 2: finite, batch-invariant, deterministic, graph == eager."""
    from cosyvoice_amd.flow import CausalMaskedDiffWithXvec
    cfg, sd = full_flow
    g = torch.Generator().manual_seed(6)
    B, n_p, n_g = 2, 250, 250
    tok = torch.randint(0, cfg.vocab_size, (B, n_g), generator=g, dtype=torch.int32)
    ptok = torch.randint(0, cfg.vocab_size, (B, n_p), generator=g, dtype=torch.int32)
    pfeat = torch.clamp(torch.randn(B, 2 * n_p, 80, generator=g) * 2 - 6, -11.5, 2.0)
    emb = torch.randn(B, cfg.spk_embed_dim, generator=g)
    flow = CausalMaskedDiffWithXvec(cfg, dtype=torch.float16).load_state_dict(sd)
    flow.encoder.static_chunk_size = 50
    m = flow.inference_batch(tok, ptok, pfeat, emb).clone()
    assert m.shape == (B, 80, 2 * n_g) and torch.isfinite(m).all()
    m2 = flow.inference_batch(tok, ptok, pfeat, emb).clone()
    assert torch.equal(m, m2)
    m1 = flow.inference_batch(tok[1:], ptok[1:], pfeat[1:], emb[1:])
    assert (m1[0] - m[1]).abs().max().item() < 2e-3  # different tile shapes -> fp16 rounding only
    flow.decoder.use_graph = True
    flow.inference_batch(tok, ptok, pfeat, emb)
    mg = flow.inference_batch(tok, ptok, pfeat, emb)
    assert torch.equal(mg, m)


@pytest.mark.parametrize("dt,tol", [(torch.bfloat16, 1.5e-1), (torch.float16, 4e-2)])
def test_full_size_llm_logp_vs_oracle(dt, tol):
    """24 layers, hidden 896, 14/2 heads, vocab 151936: teacher-forced log-probs (prefill 40 + 6 decode steps) vs the oracle."""
    from cosyvoice_amd.llm import Qwen2LM
    from oracle import llm as ol
    cfg = LlmConfig.full()
    sd = llm_state_dict(cfg)
    g = torch.Generator().manual_seed(7)
    text = torch.randint(0, cfg.vocab_size, (1, 12), generator=g, dtype=torch.int32)
    ptext = torch.randint(0, cfg.vocab_size, (1, 6), generator=g, dtype=torch.int32)
    pspeech = torch.randint(0, cfg.speech_token_size, (1, 20), generator=g, dtype=torch.int32)
    forced = torch.randint(0, cfg.speech_token_size, (6,), generator=g).tolist()
    ref = []
    list(ol.lm_inference(sd, cfg, text, ptext, pspeech, uniforms=lambda t: (0.5, 0.5), forced_tokens=forced, collect_logp=ref))
    ref = torch.stack(ref)
    lm = Qwen2LM(cfg, dtype=dt, max_batch=2, ctx_max=320, max_out=64).load_state_dict(sd)
    lp = lm.forced_logits(text, ptext, pspeech, forced).cpu()[: ref.shape[0]]
    d = (lp - ref).abs()
    top_ref, top_hip = ref.argmax(-1), lp.argmax(-1)
    print(f"full-size llm [{dt}]: logp Linf {d.max().item():.3e} mean {d.mean().item():.3e}; argmax agreement {(top_ref == top_hip).float().mean().item():.2f}")
    assert d.max().item() < tol
