"""GPU: the CosyVoice2Model-compatible orchestrator (tts / token2wav / llm_job / tts_batch) on tiny stage models."""
import pytest
import torch

from cosyvoice_amd.config import FlowConfig, HiftConfig, LlmConfig
from cosyvoice_amd.weights import flow_state_dict, hift_state_dict, llm_state_dict

pytestmark = pytest.mark.gpu


def _model(max_batch=4):
    from cosyvoice_amd.flow import CausalMaskedDiffWithXvec
    from cosyvoice_amd.hift import HiFTGenerator
    from cosyvoice_amd.llm import Qwen2LM
    from cosyvoice_amd.model import CosyVoice2Model
    lc, fc, hc = LlmConfig.tiny(), FlowConfig.tiny(), HiftConfig.tiny()
    llm = Qwen2LM(lc, dtype=torch.bfloat16, max_batch=max_batch, ctx_max=256, max_out=256)
    flow = CausalMaskedDiffWithXvec(fc, dtype=torch.float16)
    hift = HiFTGenerator(hc, dtype=torch.float32)
    m = CosyVoice2Model(llm, flow, hift, fp16=False).load_state_dicts(llm_state_dict(lc), flow_state_dict(fc), hift_state_dict(hc))
    return m, lc, fc, hc


def _inputs(lc, fc, seed=0, n_text=6, n_p=10):
    g = torch.Generator().manual_seed(seed)
    return dict(text=torch.randint(0, lc.vocab_size, (1, n_text), generator=g, dtype=torch.int32),
                prompt_text=torch.randint(0, lc.vocab_size, (1, 4), generator=g, dtype=torch.int32),
                llm_prompt_speech_token=torch.randint(0, lc.speech_token_size, (1, n_p), generator=g, dtype=torch.int32),
                flow_prompt_speech_token=torch.randint(0, lc.speech_token_size, (1, n_p), generator=g, dtype=torch.int32),
                prompt_speech_feat=torch.clamp(torch.randn(1, 2 * n_p, 80, generator=g) * 2 - 6, -11.5, 2.0),
                flow_embedding=torch.randn(1, fc.spk_embed_dim, generator=g))


def test_tts_non_stream_and_stream():
    m, lc, fc, hc = _model()
    inp = _inputs(lc, fc)
    outs = list(m.tts(**inp, stream=False))
    assert len(outs) == 1
    wav = outs[0]["tts_speech"]
    assert wav.device.type == "cpu" and wav.dim() == 2 and wav.shape[0] == 1
    assert wav.shape[1] % (2 * hc.total_upsample) == 0 and wav.shape[1] > 0   # 2 mel frames per token
    assert torch.isfinite(wav).all() and wav.abs().max().item() <= hc.audio_limit + 1e-6
    n_tok = wav.shape[1] // (2 * hc.total_upsample)
    assert 2 * 6 <= n_tok <= 20 * 6  # min/max token-text ratios (llm.py:855-856)
    assert not m.tts_speech_token_dict and not m.llm_end_dict and not m.hift_cache_dict  # per-request state cleaned up
    # streaming: chunks of token_hop_len (model.py:380-407); the tiny HiFT hop differs from 480 so only shapes are checked
    m.source_cache_len = m.mel_cache_len * hc.total_upsample
    import numpy as np
    m.speech_window = np.hamming(2 * m.source_cache_len)
    chunks = [o["tts_speech"] for o in m.tts(**inp, stream=True)]
    assert len(chunks) >= 1 and all(torch.isfinite(c).all() for c in chunks)


def test_tts_batch_matches_single_with_forced_tokens():
    m, lc, fc, hc = _model()
    B = 3
    ins = [_inputs(lc, fc, seed=s) for s in range(B)]
    g = torch.Generator().manual_seed(9)
    forced = [torch.randint(0, lc.speech_token_size, (14,), generator=g).tolist() for _ in range(B)]
    shared = ins[0]
    torch.manual_seed(0)
    wav = m.tts_batch([i["text"] for i in ins], [shared["prompt_text"]] * B, [shared["llm_prompt_speech_token"]] * B,
                      shared["flow_prompt_speech_token"].expand(B, -1), shared["prompt_speech_feat"].expand(B, -1, -1),
                      shared["flow_embedding"].expand(B, -1), forced=forced)
    assert wav.shape == (B, 14 * 2 * hc.total_upsample) and torch.isfinite(wav).all()
    # the deterministic part (flow mel) of each utterance equals its batch-1 run
    tok = torch.tensor(forced, dtype=torch.int32)
    mel_b = m.flow.inference_batch(tok, shared["flow_prompt_speech_token"].expand(B, -1), shared["prompt_speech_feat"].expand(B, -1, -1),
                                   shared["flow_embedding"].expand(B, -1)).clone()
    for b in range(B):
        mel_1 = m.flow.inference_batch(tok[b:b + 1], shared["flow_prompt_speech_token"], shared["prompt_speech_feat"], shared["flow_embedding"])
        assert (mel_1[0] - mel_b[b]).abs().max().item() < 1e-4


@pytest.mark.gpu
def test_tts_batches_cu_partition_equals_shared_streams():
    """The CU-partitioned pipeline (decode loop and flow/HiFT on disjoint CU-masked streams, graphs replayed launch by
    launch from two host threads; optionally several batches per token loop) produces bit-identical waveforms to the
    two-plain-streams pipeline."""
    m, lc, fc, hc = _model()
    B, nb = 2, 5
    shared = _inputs(lc, fc, seed=0)
    g = torch.Generator().manual_seed(11)

    def batches():
        out = []
        for i in range(nb):
            ins = [_inputs(lc, fc, seed=10 * i + s) for s in range(B)]
            forced = [torch.randint(0, lc.speech_token_size, (14,), generator=torch.Generator().manual_seed(100 + i)).tolist()] * B
            out.append(dict(texts=[x["text"].cuda() for x in ins], prompt_texts=[shared["prompt_text"].cuda()] * B,
                            llm_prompt_speech_tokens=[shared["llm_prompt_speech_token"].cuda()] * B,
                            flow_prompt_speech_tokens=shared["flow_prompt_speech_token"].cuda().expand(B, -1),
                            prompt_speech_feats=shared["prompt_speech_feat"].cuda().expand(B, -1, -1),
                            flow_embeddings=shared["flow_embedding"].cuda().expand(B, -1), forced=forced))
        return out

    torch.manual_seed(0)
    ref = [w.clone() for w in m.tts_batches(batches(), llm_cu_slots=0)]
    # merge: consecutive batches decoded by ONE token loop (2 x 2 rows <= max_batch 4; 3 is clipped to what fits)
    for k, loops, to_host, merge in ((12, 1, True, 1), (4, 2, False, 1), (8, 3, True, 1), (8, 2, True, 2), (8, 1, False, 3)):
        torch.manual_seed(0)
        m.llm_merge = merge
        got = [w.cpu().clone() for w in m.tts_batches(batches(), to_host=to_host, llm_cu_slots=k, llm_loops=loops)]
        m.llm_merge = 1
        assert len(got) == nb
        for a, b in zip(ref, got):
            assert a.shape == b.shape and torch.isfinite(b).all()
            assert torch.equal(a, b)


@pytest.mark.gpu
def test_tts_batch_free_running_ragged_lengths():
    """Without teacher forcing every utterance (different text lengths) stops at its own step: tts_batch returns one
    waveform per utterance, 2 * hop * n_tokens samples each."""
    m, lc, fc, hc = _model()
    ins = [_inputs(lc, fc, seed=s, n_text=n) for s, n in ((1, 3), (2, 6), (3, 4))]
    shared = ins[0]
    B = len(ins)
    torch.manual_seed(0)
    wavs = m.tts_batch([i["text"] for i in ins], [shared["prompt_text"]] * B, [shared["llm_prompt_speech_token"]] * B,
                       shared["flow_prompt_speech_token"].expand(B, -1), shared["prompt_speech_feat"].expand(B, -1, -1),
                       shared["flow_embedding"].expand(B, -1))
    wavs = wavs if isinstance(wavs, list) else list(wavs)
    assert len(wavs) == B
    for w, i in zip(wavs, ins):
        n = w.numel() // (2 * hc.total_upsample)
        assert w.numel() == n * 2 * hc.total_upsample and 2 * i["text"].shape[1] <= n <= 20 * i["text"].shape[1]
        assert torch.isfinite(w).all() and w.abs().max() <= 0.99 + 1e-6


@pytest.mark.gpu
def test_v1_wiring_orchestrator_chunk_schedule(golden_dir):
    """CosyVoiceModel (the v1 wiring the fork drives CosyVoice2 modules with): vc() streaming and not, against the chunk
    lengths the reference's own CosyVoiceModel yields on the same inputs (golden); the non-stream waveform equals flow + HiFT
    called directly."""
    import os
    import numpy as np
    from cosyvoice_amd.flow import CausalMaskedDiffWithXvec
    from cosyvoice_amd.hift import HiFTGenerator
    from cosyvoice_amd.model import CosyVoiceModel
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, "v1_orchestrator.npz")).items()}
    fc, hc = FlowConfig.tiny(), HiftConfig.v1()
    flow = CausalMaskedDiffWithXvec(fc, dtype=torch.float16).load_state_dict(flow_state_dict(fc))
    hift = HiFTGenerator(hc, dtype=torch.float32).load_state_dict(hift_state_dict(hc))

    class _NoLLM:
        fp16 = False

    m = CosyVoiceModel(_NoLLM(), flow, hift, fp16=False, sr=22050)
    assert m.mel_overlap_len == 68 and m.token_min_hop_len == 50 and flow.decoder.estimator.static_chunk_size == 0
    args = (g["source_speech_token"], g["prompt_token"], g["prompt_feat"], g["embedding"])
    chunks = [o["tts_speech"] for o in m.vc(*args, stream=True)]
    assert [c.shape[1] for c in chunks] == g["stream_chunk_samples"].tolist()
    assert all(torch.isfinite(c).all() and c.abs().max() <= 0.99 + 1e-6 for c in chunks)
    full = [o["tts_speech"] for o in m.vc(*args, stream=False)]
    assert [c.shape[1] for c in full] == g["full_samples"].tolist()
    assert not m.tts_speech_token_dict and not m.hift_cache_dict and not m.mel_overlap_dict   # per-request state released
    # speed change (non-stream only, model.py:165-168): linear interpolation of the mel
    fast = [o["tts_speech"] for o in m.vc(*args, stream=False, speed=2.0)]
    assert fast[0].shape[1] == int(g["full_samples"][0]) // 2


@pytest.mark.gpu
@pytest.mark.parametrize("fm", [1, 2])
def test_tts_batches_merged_loops_free_running(fm):
    """Two batches per token loop without teacher forcing: every utterance ends at its own step, the merged job's tokens are
    handed back per batch, flow + HiFT run per batch (fm = 1) or per merged group of two batches (fm = 2: the ragged flow pass over the
    concatenated conditioning, waveforms split back per batch) over ragged lengths."""
    m, lc, fc, hc = _model()
    m.flow_merge = fm
    B, nb = 2, 3
    shared = _inputs(lc, fc, seed=0)
    batches = []
    for i in range(nb):
        ins = [_inputs(lc, fc, seed=20 * i + s, n_text=3 + s + i) for s in range(B)]
        batches.append(dict(texts=[x["text"].cuda() for x in ins], prompt_texts=[shared["prompt_text"].cuda()] * B,
                            llm_prompt_speech_tokens=[shared["llm_prompt_speech_token"].cuda()] * B,
                            flow_prompt_speech_tokens=shared["flow_prompt_speech_token"].cuda().expand(B, -1),
                            prompt_speech_feats=shared["prompt_speech_feat"].cuda().expand(B, -1, -1),
                            flow_embeddings=shared["flow_embedding"].cuda().expand(B, -1)))
    m.llm_merge = 2
    try:
        outs = list(m.tts_batches(batches, to_host=True, llm_cu_slots=8, llm_loops=2))
    finally:
        m.llm_merge = 1
    assert len(outs) == nb
    for i, wavs in enumerate(outs):
        wavs = wavs if isinstance(wavs, list) else list(wavs)
        assert len(wavs) == B
        for s, w in enumerate(wavs):
            n_text = 3 + s + i
            n = w.numel() // (2 * hc.total_upsample)
            assert w.numel() == n * 2 * hc.total_upsample and 2 * n_text <= n <= 20 * n_text
            assert torch.isfinite(w).all() and w.abs().max() <= 0.99 + 1e-6


@pytest.mark.gpu
def test_concurrent_tts_requests_overlap_and_equal_solo_runs():
    """Request threads on ONE model object, as the reference interleaves them (cli/model.py:62,119,189: one LLM thread + side
    stream per tts() call): every request decodes on its own context (Qwen2LM.new_context), so the token loops run at the same
    time, and a request seeded with ``seed=`` produces exactly the waveform it produces alone."""
    import threading
    m, lc, fc, hc = _model()
    reqs = [dict(_inputs(lc, fc, seed=s, n_text=n), seed=100 + s) for s, n in ((1, 9), (2, 11), (3, 10))]
    # the vocoder's source noise comes from torch's global GPU generator, so waveforms are not comparable across runs: compare
    # the token sequence every request hands to token2wav (recorded per calling thread) and the waveform length
    seen, orig = {}, m.token2wav

    def rec(*a, **kw):
        seen[threading.get_ident()] = kw["token"].clone()
        return orig(*a, **kw)
    m.token2wav = rec

    def one(r):
        n = sum(o["tts_speech"].shape[1] for o in m.tts(**r, stream=False))
        return seen[threading.get_ident()].flatten().tolist(), n
    alone = [one(r) for r in reqs]
    got, errs = [None] * len(reqs), []
    m._llm_spans.clear()
    go = threading.Barrier(len(reqs))

    def run(i):
        try:
            go.wait()
            got[i] = one(reqs[i])
        except Exception as e:      # surfaced in the main thread
            errs.append(e)

    ths = [threading.Thread(target=run, args=(i,)) for i in range(len(reqs))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    assert got == alone
    assert all(len(t) >= 18 for t, _ in got)
    sp = sorted(v for v in m._llm_spans.values() if v[1] is not None)
    assert len(sp) == len(reqs)
    # the token loops overlap in time: the second one starts before the first one ends
    assert sp[1][0] < sp[0][1], sp
    assert not m.tts_speech_token_dict and not m.llm_end_dict and not m.hift_cache_dict


@pytest.mark.gpu
def test_free_running_requests_differ_and_manual_seed_reproduces():
    """The sampler's Philox key carries a per-request nonce drawn from torch's global generator: consecutive requests draw
    different uniforms (the decode graph is NOT recaptured), torch.manual_seed reproduces a sequence of requests."""
    m, lc, fc, hc = _model()
    r = _inputs(lc, fc, seed=5, n_text=10)
    def toks():
        out = []
        m.tts_speech_token_dict["x"], m.llm_end_dict["x"] = out, False
        m.llm_job(r["text"], r["prompt_text"], r["llm_prompt_speech_token"], torch.zeros(0, 192), "x")
        m.tts_speech_token_dict.pop("x"); m.llm_end_dict.pop("x")
        return list(out)
    torch.manual_seed(7)
    a1, a2 = toks(), toks()
    n_graphs = len(m.llm._graphs)
    torch.manual_seed(7)
    b1, b2 = toks(), toks()
    assert a1 == b1 and a2 == b2
    assert a1 != a2
    assert len(m.llm._graphs) == n_graphs


def test_tts_llm_thread_failure_reaches_the_caller():
    """An exception inside the request's LLM thread must end the consumer loop (llm_end_dict is set in a finally) and be re-raised by
    tts() in the calling thread — streaming and not — with the per-request state cleaned up and the decode context back in the pool."""
    m, lc, fc, hc = _model()

    class Boom(RuntimeError):
        pass

    orig = m.llm.new_context

    def failing_context():
        ctx = orig()

        def inference(**kw):
            yield 3
            raise Boom("decode failed")
        ctx.inference = inference
        return ctx
    m.llm.new_context = failing_context
    for stream in (False, True):
        with pytest.raises(Boom):
            list(m.tts(**_inputs(lc, fc), stream=stream))
        assert not m.tts_speech_token_dict and not m.llm_end_dict and not m.hift_cache_dict and not m._llm_errors
    assert len(m._req_pool) == 1          # the one context made so far was released both times
    assert all(c is not m.llm for c, _ in m._req_pool)   # self.llm stays reserved for tts_batch / tts_batches


def test_reload_invalidates_descriptors_and_graphs():
    """load_state_dict on stages that have already run: the HiFT decode descriptor (raw pointers of every conv tensor), the captured
    decode-step graphs and the captured Euler loops are rebuilt — results after a second load equal a freshly built model's."""
    from cosyvoice_amd.flow import CausalMaskedDiffWithXvec
    from cosyvoice_amd.hift import HiFTGenerator
    from cosyvoice_amd.llm import Qwen2LM
    lc, fc, hc = LlmConfig.tiny(), FlowConfig.tiny(), HiftConfig.tiny()
    g = torch.Generator().manual_seed(4)
    mel = torch.clamp(torch.randn(1, 80, 20, generator=g) * 2 - 6, -11.5, 2.0).cuda()
    s = (torch.randn(1, 1, 20 * hc.total_upsample, generator=g) * 0.05).cuda()
    sd_a, sd_b = hift_state_dict(hc, seed=1986), hift_state_dict(hc, seed=7)
    h = HiFTGenerator(hc, dtype=torch.float32).load_state_dict(sd_a)
    w_a = h.decode(mel, s).clone()
    h.load_state_dict(sd_b)
    w_b = h.decode(mel, s).clone()
    w_ref = HiFTGenerator(hc, dtype=torch.float32).load_state_dict(sd_b).decode(mel, s).clone()
    assert torch.equal(w_b, w_ref) and not torch.equal(w_a, w_b)
    # LLM: teacher-forced log-probs through the captured step graph
    text = torch.randint(0, lc.vocab_size, (1, 5), generator=g, dtype=torch.int32)
    ptext = torch.randint(0, lc.vocab_size, (1, 3), generator=g, dtype=torch.int32)
    pspeech = torch.randint(0, lc.speech_token_size, (1, 8), generator=g, dtype=torch.int32)
    forced = torch.randint(0, lc.speech_token_size, (6,), generator=g).tolist()
    la, lb = llm_state_dict(lc, seed=1986), llm_state_dict(lc, seed=11)
    lm = Qwen2LM(lc, dtype=torch.float16, max_batch=2, ctx_max=128, max_out=64).load_state_dict(la)
    p_a = lm.forced_logits(text, ptext, pspeech, forced).clone()
    lm.load_state_dict(lb)
    p_b = lm.forced_logits(text, ptext, pspeech, forced).clone()
    p_ref = Qwen2LM(lc, dtype=torch.float16, max_batch=2, ctx_max=128, max_out=64).load_state_dict(lb).forced_logits(text, ptext, pspeech, forced)
    assert torch.equal(p_b, p_ref) and not torch.equal(p_a, p_b)
    # flow: the captured Euler loop
    tok = torch.randint(0, fc.vocab_size, (1, 12), generator=g, dtype=torch.int32)
    ptok = torch.randint(0, fc.vocab_size, (1, 6), generator=g, dtype=torch.int32)
    pfeat = torch.clamp(torch.randn(1, 12, 80, generator=g) * 2 - 6, -11.5, 2.0)
    emb = torch.randn(1, fc.spk_embed_dim, generator=g)
    fa, fb = flow_state_dict(fc, seed=1986), flow_state_dict(fc, seed=13)
    fl = CausalMaskedDiffWithXvec(fc, dtype=torch.float16).load_state_dict(fa)
    fl.decoder.use_graph = True
    for _ in range(2):
        m_a = fl.inference_batch(tok, ptok, pfeat, emb).clone()
    fl.load_state_dict(fb)
    for _ in range(2):
        m_b = fl.inference_batch(tok, ptok, pfeat, emb).clone()
    fr = CausalMaskedDiffWithXvec(fc, dtype=torch.float16).load_state_dict(fb)
    fr.decoder.use_graph = True
    for _ in range(2):
        m_ref = fr.inference_batch(tok, ptok, pfeat, emb).clone()
    assert torch.equal(m_b, m_ref) and not torch.equal(m_a, m_b)


@pytest.mark.parametrize("cu_slots", [8, 0])
def test_tts_batches_conditioning_slots_lifecycle(cu_slots):
    """Batches drawn lazily from a generator, each with its own conditioning slot (cosyvoice_amd.dist.ConditioningRing): on_start runs in
    batch order when the pipeline admits the batch, on_done once its waveform has been collected, a slot is never handed out twice at once,
    and every batch's audio is what that batch's OWN conditioning gives (distinct prompt per batch) — CU-partitioned and shared-CU paths."""
    from cosyvoice_amd import dist as cd
    m, lc, fc, hc = _model()
    m.llm_merge = 1
    n_batches, B, n_p = 10, 2, 10        # more batches than slots: slots are recycled
    g = torch.Generator().manual_seed(21)
    texts = [torch.randint(0, lc.vocab_size, (1, 6), generator=g, dtype=torch.int32).cuda() for _ in range(B)]
    forced = [torch.randint(0, lc.speech_token_size, (12,), generator=g).tolist() for _ in range(B)]
    conds = []
    for i in range(n_batches):
        pf = torch.clamp(torch.randn(1, 2 * n_p, 80, generator=g) * 2 - 6, -11.5, 2.0)
        em = torch.randn(1, fc.spk_embed_dim, generator=g)
        ps = torch.randint(0, lc.speech_token_size, (1, n_p), generator=g, dtype=torch.int32)
        pt = torch.randint(0, lc.vocab_size, (1, 4), generator=g, dtype=torch.int32)
        conds.append((pf, em, ps, pt))
    payloads = [cd.pack_conditioning(*c)[0].cuda() for c in conds]
    _, layout = cd.pack_conditioning(*conds[0])
    ring = cd.ConditioningRing(7, layout, torch.device("cuda"))   # (llm_loops + 2) * llm_merge + llm_merge + 2: the pipeline's depth
    events, live = [], set()

    def batches():
        for i in range(n_batches):
            slot = ring.acquire()
            assert slot not in live
            live.add(slot)

            def start(i=i, slot=slot):
                events.append(("start", i))
                ring.slots[slot].copy_(payloads[i])        # stands in for the broadcast into this batch's slot
                ring.after_broadcast(slot)

            def done(i=i, slot=slot):
                events.append(("done", i))
                live.discard(slot)
                ring.release(slot)
            pf, em, ps, pt = ring.tensors(slot)
            yield dict(texts=texts, prompt_texts=[pt] * B, llm_prompt_speech_tokens=[ps] * B, flow_prompt_speech_tokens=ps.expand(B, -1),
                       prompt_speech_feats=pf.expand(B, -1, -1), flow_embeddings=em.expand(B, -1), forced=forced, on_start=start, on_done=done)
    torch.manual_seed(0)
    wavs = [w.clone() for w in m.tts_batches(batches(), to_host=True, llm_cu_slots=cu_slots, llm_loops=2)]
    assert len(wavs) == n_batches and ring.in_use() == 0 and ring.high_water <= 7
    starts = [i for k, i in events if k == "start"]
    dones = [i for k, i in events if k == "done"]
    assert starts == list(range(n_batches)) and dones == list(range(n_batches))
    for i in range(n_batches):
        assert events.index(("start", i)) < events.index(("done", i))
    # a batch's waveform length / content follows ITS prompt: the mel (deterministic) differs between batches with different prompts;
    # compare each batch against a solo run of the same conditioning through tts_batch (vocoder noise differs: compare lengths + energy)
    for i in (0, 4, 9):
        pf, em, ps, pt = (t.cuda() for t in conds[i])
        solo = m.tts_batch(texts, [pt] * B, [ps] * B, ps.expand(B, -1), pf.expand(B, -1, -1), em.expand(B, -1), forced=forced)
        assert solo.shape == wavs[i].shape
        assert (solo.abs().mean() - wavs[i].abs().mean()).abs().item() < 0.02 * max(1e-3, solo.abs().mean().item()) + 5e-3
    m.close()


@pytest.mark.parametrize("fm", [2, 3])
def test_tts_batches_flow_merge_matches_unmerged(fm):
    """``flow_merge``: consecutive batches of one decode job share ONE flow + HiFT pass (the row-block kernels fill the flow CUs in whole
    rounds of workgroups, so 16 / 24 utterances cost less per utterance than 8).  Every utterance's mel must be bit-identical to the
    unmerged run (the flow is batch-invariant), each batch must get ITS rows back in order, and on_done must follow the batch's result."""
    import hashlib
    m, lc, fc, hc = _model(max_batch=8)      # 3 batches of 2 rows per decode job
    n_batches, B, n_p = 9, 2, 10
    g = torch.Generator().manual_seed(33)
    texts = [torch.randint(0, lc.vocab_size, (1, 6), generator=g, dtype=torch.int32).cuda() for _ in range(B)]
    conds, forced = [], []
    for i in range(n_batches):
        pf = torch.clamp(torch.randn(1, 2 * n_p, 80, generator=g) * 2 - 6, -11.5, 2.0).cuda()
        em = torch.randn(1, fc.spk_embed_dim, generator=g).cuda()
        ps = torch.randint(0, lc.speech_token_size, (1, n_p), generator=g, dtype=torch.int32).cuda()
        pt = torch.randint(0, lc.vocab_size, (1, 4), generator=g, dtype=torch.int32).cuda()
        conds.append((pf, em, ps, pt))
        forced.append([torch.randint(0, lc.speech_token_size, (12,), generator=g).tolist() for _ in range(B)])   # own tokens per batch

    def run(merge):
        m.llm_merge, m.flow_merge = 3, merge
        digests, done = [], []
        orig = m.flow.inference_batch

        def rec(tok, *a, **k):
            mel = orig(tok, *a, **k)
            for j in range(mel.shape[0]):
                digests.append((tuple(tok[j].tolist()), hashlib.sha256(mel[j].float().cpu().numpy().tobytes()).hexdigest()))
            return mel
        m.flow.inference_batch = rec
        try:
            def batches():
                for i, (pf, em, ps, pt) in enumerate(conds):
                    yield dict(texts=texts, prompt_texts=[pt] * B, llm_prompt_speech_tokens=[ps] * B, flow_prompt_speech_tokens=ps.expand(B, -1),
                               prompt_speech_feats=pf.expand(B, -1, -1), flow_embeddings=em.expand(B, -1), forced=forced[i],
                               on_done=lambda i=i: done.append(i))
            wavs = [w.clone() for w in m.tts_batches(batches(), to_host=True, llm_cu_slots=8, llm_loops=2)]
        finally:
            m.flow.inference_batch = orig
        return wavs, dict(digests), done, [len(d) for d in [digests]]
    w1, d1, done1, _ = run(1)
    wm, dm, donem, _ = run(fm)
    assert done1 == list(range(n_batches)) and donem == list(range(n_batches))
    assert len(d1) == n_batches * B and d1 == dm            # keyed by the utterance's own token list: same mel bits, merged or not
    assert len(wm) == n_batches
    for a, b in zip(w1, wm):
        assert a.shape == b.shape and torch.isfinite(b).all()
        assert (a.abs().mean() - b.abs().mean()).abs().item() < 0.02 * max(1e-3, a.abs().mean().item()) + 5e-3   # vocoder noise differs
    m.close()
