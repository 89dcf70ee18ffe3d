"""Orchestrator — drop-in for the reference's ``CosyVoice2Model`` (/root/reference/cosyvoice/cli/model.py:295-424):
per-request uuid state, LLM producer thread (``llm_job``), ``token2wav`` (flow -> HiFT with the streaming
mel/source/speech caches and hamming cross-fade), ``tts`` generator yielding ``{'tts_speech': Tensor(1,S) cpu}``.

The stage objects are the HIP-backed ``cosyvoice_amd.llm.Qwen2LM`` / ``flow.CausalMaskedDiffWithXvec`` /
``hift.HiFTGenerator`` (they expose what the orchestrator touches, SURVEY.md §8b level 1), so
``cosyvoice.cli`` code that builds a CosyVoice2Model keeps working with this class.

``tts_batch`` is the build's utterance-batched entry (the reference is batch-1): LLM decode batched over B
sequences, flow batched over B CFG pairs, HiFT batched over B mels.
"""
import os
import collections
import threading
import time
import uuid
from typing import List, Optional

import numpy as np
import torch
import torch.nn.functional as F


def fade_in_out(fade_in_mel, fade_out_mel, window):
    """utils/common.py:149-157 (host-side cross-fade of the stream seam)."""
    device = fade_in_mel.device
    fade_in_mel, fade_out_mel = fade_in_mel.cpu().clone(), fade_out_mel.cpu()
    n = int(window.shape[0] / 2)
    w = torch.as_tensor(window)   # float64 hamming window, as the reference's numpy broadcast (result is cast back on assignment)
    fade_in_mel[..., :n] = fade_in_mel[..., :n] * w[:n] + fade_out_mel[..., -n:] * w[n:]
    return fade_in_mel.to(device)


class CosyVoice2Model:
    def __init__(self, llm, flow, hift, fp16: bool = False):
        self.device = torch.device("cuda")
        if not torch.cuda.is_available():
            raise RuntimeError("cosyvoice_amd needs an MI355X (no CPU fallback)")
        self.llm, self.flow, self.hift = llm, flow, hift
        self.fp16 = fp16
        self.llm.fp16 = fp16
        self.flow.fp16 = fp16
        self.token_hop_len = 2 * self.flow.input_frame_rate
        # model.py:313-315
        if hasattr(self.flow, "token_mel_ratio"):     # CosyVoice2 flow; the v1 MaskedDiffWithXvec has no chunk attention
            self.flow.encoder.static_chunk_size = 2 * self.flow.input_frame_rate
            self.flow.decoder.estimator.static_chunk_size = 2 * self.flow.input_frame_rate * self.flow.token_mel_ratio
        self.mel_cache_len = 8
        self.source_cache_len = int(self.mel_cache_len * 480)
        self.speech_window = np.hamming(2 * self.source_cache_len)
        self.stream_scale_factor = 1
        # high-priority side stream: the decode loop is a chain of short dependent kernels that must not queue behind the
        # flow GEMMs of the previous batch when both streams are busy (tts_batches)
        self.llm_context = torch.cuda.stream(torch.cuda.Stream(self.device, priority=-1))
        # streaming requests run the flow at the next multiple of this many tokens (tail masked, exact): requests of arbitrary
        # prompt / chunk lengths then share a handful of captured estimator graphs instead of capturing one per length
        self.stream_length_bucket = 25
        self.llm_cu_slots = 8   # tts_batches: CU slots per XCD (of 32) owned by the decode loops; 0 = no partition
        self.llm_loops = 2      # tts_batches: concurrent decode loops (each its own batch / KV caches) on those CUs
        self.llm_merge = 1      # tts_batches: consecutive batches decoded by ONE token loop (rows <= llm.max_batch)
        self.flow_merge = 1     # tts_batches: up to this many consecutive batches whose tokens are ready share ONE flow + HiFT pass
        self.llm_ramp = (1, 2, 2, 3)   # tts_batches: batches per decode job while the pipeline fills (then llm_merge); CV_LLM_RAMP overrides
        self.lock = threading.Lock()
        # The reference's modules are re-entrant (every call builds its own activations / KV cache) and it interleaves requests:
        # one LLM thread + side stream per tts() call (cli/model.py:62,119,189).  Here a decode loop owns KV caches, device state and
        # captured graphs, so every request takes its OWN decode context over the one set of weights (Qwen2LM.new_context) from a
        # pool of ``max_llm_requests`` and runs it on that context's stream: the token loops of concurrent requests overlap.  The
        # flow / vocoder objects keep one workspace set: token2wav calls take turns (flow_lock) — they are GPU-bound, not
        # latency-bound, so nothing is gained by interleaving them (batches go through tts_batch / tts_batches instead).
        self.max_llm_requests = 4
        self.llm_lock = threading.Lock()          # only for LMs without decode contexts (the v1 TransformerLM)
        self._req_pool_lock = threading.Condition()
        self._req_pool = None                     # [(context, stream)] free for requests
        self._req_made = 0
        self.flow_lock = threading.Lock()
        self.tts_speech_token_dict = {}
        self.llm_end_dict = {}
        self.hift_cache_dict = {}
        self._llm_spans = collections.OrderedDict()   # uuid -> [start, end] wall-clock of the request's token loop (diagnostics / tests); the last 64 requests
        self._llm_errors = {}     # uuid -> exception raised inside the request's LLM thread; tts() re-raises it in the caller
        self.overlap_hift = os.environ.get("CV_OVERLAP_HIFT", "0") != "0"   # tts_batches: HiFT of batch i on its own stream beside flow of batch i + 1 (measured SLOWER: 455 vs 548 audio-s/s — the vocoder GEMMs and the flow chain disturb each other on shared CUs; kept for experiments)
        # tts_batches: where the decode jobs' prefills run.  True = on the flow CUs (throughput-bound GEMM work over B x 282 rows); False = on
        # the decode CUs with the token loop.  Re-checked in round 2, when the flow + HiFT stream became the bottleneck: still better on the
        # flow CUs (622-631 vs 605-612 audio-s/s) — on 64 CUs a 32-row prefill stalls both token loops for ~0.1 s per job
        self.prefill_on_flow_cus = os.environ.get("CV_PREFILL_ON_FLOW", "1") != "0"
        self.pipeline_stats = None  # set to a list to collect (stage, batches, start, end) of every tts_batches job

    def _drop_contexts(self):
        """Forget every extra decode context (they share the weight tensors of the LM they were made from: after a reload they
        would keep decoding with the old weights) and their captured graphs."""
        from . import ops
        with self._req_pool_lock:
            ctxs = [c for c, _ in (self._req_pool or [])] + [c for c in (getattr(self, "_llm_contexts", None) or []) if c is not self.llm]
            for c in ctxs:
                ops.drop_graphs(getattr(c, "_graphs", {}))
            self._req_pool, self._req_made = None, 0
            self._llm_contexts = None

    def load(self, llm_model, flow_model, hift_model):
        """model.py:71-81 — three flat state-dict files with the reference key names."""
        self._drop_contexts()
        self.llm.load_state_dict(torch.load(llm_model, map_location="cpu", weights_only=True), strict=False)
        self.flow.load_state_dict(torch.load(flow_model, map_location="cpu", weights_only=True), strict=False)
        hift_sd = {k.replace("generator.", ""): v for k, v in torch.load(hift_model, map_location="cpu", weights_only=True).items()}
        self.hift.load_state_dict(hift_sd, strict=False)

    def load_state_dicts(self, llm_sd, flow_sd, hift_sd):
        self._drop_contexts()
        self.llm.load_state_dict(llm_sd)
        self.flow.load_state_dict(flow_sd)
        self.hift.load_state_dict(hift_sd)
        return self

    def _acquire_llm(self):
        """A free (decode context, stream) pair for one request; grows the pool up to max_llm_requests, then waits.  Every pooled
        context is a Qwen2LM.new_context(): ``self.llm`` itself stays reserved for the batch paths (tts_batch / tts_batches use it
        without taking the pool), so a tts() request never shares KV caches or device state with a running batch."""
        with self._req_pool_lock:
            if self._req_pool is None:
                self._req_pool, self._req_made = [], 0
            while True:
                if self._req_pool:
                    return self._req_pool.pop()
                if self._req_made < self.max_llm_requests:
                    self._req_made += 1
                    break
                self._req_pool_lock.wait()
        return self.llm.new_context(), torch.cuda.Stream(self.device, priority=-1)

    def _release_llm(self, pair):
        with self._req_pool_lock:
            self._req_pool.append(pair)
            self._req_pool_lock.notify()

    def llm_job(self, text, prompt_text, llm_prompt_speech_token, llm_embedding, uuid_, seed=None):
        # model.py:116-128.  The fork's phoneme LMs (Qwen2LM_Phoneme_Src2.inference, llm.py:1687-1699) take
        # text = (bpe ids, phoneme factors) and matching length tuples: tuples are passed through in that form.
        dev = self.device
        if isinstance(text, (tuple, list)):
            to = lambda t: tuple(x.to(dev) for x in t)
            ln = lambda t: tuple(torch.tensor([x.shape[1]], dtype=torch.int32) for x in t)
            text_a, text_len, ptext_a, ptext_len = to(text), ln(text), to(prompt_text), ln(prompt_text)
        else:
            text_a, text_len = text.to(dev), torch.tensor([text.shape[1]], dtype=torch.int32)
            ptext_a, ptext_len = prompt_text.to(dev), torch.tensor([prompt_text.shape[1]], dtype=torch.int32)
        kw = dict(text=text_a, text_len=text_len, prompt_text=ptext_a, prompt_text_len=ptext_len,
                  prompt_speech_token=llm_prompt_speech_token.to(self.device),
                  prompt_speech_token_len=torch.tensor([llm_prompt_speech_token.shape[1]], dtype=torch.int32), embedding=llm_embedding)
        try:
            if hasattr(self.llm, "new_context"):
                pair = self._acquire_llm()
                try:
                    ctx, stream = pair
                    with self.lock:
                        self._llm_spans[uuid_] = [time.perf_counter(), None]
                        while len(self._llm_spans) > 64:
                            self._llm_spans.popitem(last=False)
                    span = self._llm_spans[uuid_]
                    with torch.cuda.stream(stream):
                        for i in ctx.inference(nonce=seed, **kw):
                            self.tts_speech_token_dict[uuid_].append(i)
                        stream.synchronize()
                    span[1] = time.perf_counter()
                finally:
                    self._release_llm(pair)
            else:
                with self.llm_lock, self.llm_context:
                    for i in self.llm.inference(**kw):
                        self.tts_speech_token_dict[uuid_].append(i)
        except BaseException as e:      # the consumer loop of tts() polls llm_end_dict: it must see the end, and the error
            self._llm_errors[uuid_] = e
        finally:
            self.llm_end_dict[uuid_] = True

    def token2wav(self, *args, **kwargs):
        """Reference signature (model.py:334 / :130 for the v1 wiring); one call at a time per model object (flow_lock)."""
        with self.flow_lock:
            return self._token2wav(*args, **kwargs)

    def _token2wav(self, token, prompt_token, prompt_feat, embedding, uuid_, token_offset, finalize=False, speed=1.0):
        # model.py:334-366
        tts_mel, _ = self.flow.inference(token=token.to(self.device),
                                         token_len=torch.tensor([token.shape[1]], dtype=torch.int32),
                                         prompt_token=prompt_token.to(self.device),
                                         prompt_token_len=torch.tensor([prompt_token.shape[1]], dtype=torch.int32),
                                         prompt_feat=prompt_feat.to(self.device),
                                         prompt_feat_len=torch.tensor([prompt_feat.shape[1]], dtype=torch.int32),
                                         embedding=embedding.to(self.device), finalize=finalize,
                                         # a streaming request's chunk calls share its chunk-causal encoder state
                                         **({"cache_key": uuid_} if (getattr(self.flow, "supports_stream_cache", False)
                                                                     and (not finalize or token_offset > 0)) else {}))
        tts_mel = tts_mel[:, :, token_offset * self.flow.token_mel_ratio:]
        if self.hift_cache_dict[uuid_] is not None:
            hift_cache_mel, hift_cache_source = self.hift_cache_dict[uuid_]["mel"], self.hift_cache_dict[uuid_]["source"]
            tts_mel = torch.concat([hift_cache_mel, tts_mel], dim=2)
        else:
            hift_cache_source = torch.zeros(1, 1, 0)
        if finalize is False:
            tts_speech, tts_source = self.hift.inference(speech_feat=tts_mel, cache_source=hift_cache_source)
            tts_speech, tts_source = tts_speech.clone(), tts_source.clone()
            if self.hift_cache_dict[uuid_] is not None:
                tts_speech = fade_in_out(tts_speech, self.hift_cache_dict[uuid_]["speech"], self.speech_window)
            self.hift_cache_dict[uuid_] = {"mel": tts_mel[:, :, -self.mel_cache_len:].clone(),
                                           "source": tts_source[:, :, -self.source_cache_len:].clone(),
                                           "speech": tts_speech[:, -self.source_cache_len:].clone()}
            tts_speech = tts_speech[:, :-self.source_cache_len]
        else:
            if speed != 1.0:
                assert self.hift_cache_dict[uuid_] is None, "speed change only support non-stream inference mode"
                tts_mel = F.interpolate(tts_mel, size=int(tts_mel.shape[2] / speed), mode="linear")
            tts_speech, tts_source = self.hift.inference(speech_feat=tts_mel, cache_source=hift_cache_source)
            tts_speech = tts_speech.clone()
            if self.hift_cache_dict[uuid_] is not None:
                tts_speech = fade_in_out(tts_speech, self.hift_cache_dict[uuid_]["speech"], self.speech_window)
        return tts_speech

    def tts(self, text, flow_embedding, llm_embedding=torch.zeros(0, 192), prompt_text=torch.zeros(1, 0, dtype=torch.int32),
            llm_prompt_speech_token=torch.zeros(1, 0, dtype=torch.int32), flow_prompt_speech_token=torch.zeros(1, 0, dtype=torch.int32),
            prompt_speech_feat=torch.zeros(1, 0, 80), stream=False, speed=1.0, **kwargs):
        # model.py:368-424
        this_uuid = str(uuid.uuid1())
        with self.lock:
            self.tts_speech_token_dict[this_uuid], self.llm_end_dict[this_uuid] = [], False
            self.hift_cache_dict[this_uuid] = None
        # ``seed`` (extra keyword, absent from the reference): Philox key material of this request's sampler; None draws it from
        # torch's global generator (as the reference samples from the global RNG), an int makes the request reproducible
        p = threading.Thread(target=self.llm_job, args=(text, prompt_text, llm_prompt_speech_token, llm_embedding, this_uuid, kwargs.get("seed")))
        p.start()

        def check_llm():
            err = self._llm_errors.pop(this_uuid, None)
            if err is not None:
                raise err

        try:
            if stream is True:
                self.flow.length_bucket = self.stream_length_bucket
                token_offset = 0
                need = self.token_hop_len + self.flow.pre_lookahead_len
                while True:
                    time.sleep(0.02)
                    if len(self.tts_speech_token_dict[this_uuid]) - token_offset >= need:
                        this_tok = torch.tensor(self.tts_speech_token_dict[this_uuid][:token_offset + need]).unsqueeze(dim=0)
                        this_speech = self.token2wav(token=this_tok, prompt_token=flow_prompt_speech_token, prompt_feat=prompt_speech_feat,
                                                     embedding=flow_embedding, uuid_=this_uuid, token_offset=token_offset, finalize=False)
                        token_offset += self.token_hop_len
                        yield {"tts_speech": this_speech.cpu()}
                    if self.llm_end_dict[this_uuid] is True and len(self.tts_speech_token_dict[this_uuid]) - token_offset < need:
                        break
                p.join()
                check_llm()
                this_tok = torch.tensor(self.tts_speech_token_dict[this_uuid]).unsqueeze(dim=0)
                this_speech = self.token2wav(token=this_tok, prompt_token=flow_prompt_speech_token, prompt_feat=prompt_speech_feat,
                                             embedding=flow_embedding, uuid_=this_uuid, token_offset=token_offset, finalize=True)
                yield {"tts_speech": this_speech.cpu()}
            else:
                p.join()
                check_llm()
                this_tok = torch.tensor(self.tts_speech_token_dict[this_uuid]).unsqueeze(dim=0)
                this_speech = self.token2wav(token=this_tok, prompt_token=flow_prompt_speech_token, prompt_feat=prompt_speech_feat,
                                             embedding=flow_embedding, uuid_=this_uuid, token_offset=0, finalize=True, speed=speed)
                yield {"tts_speech": this_speech.cpu()}
        finally:
            p.join()
            with self.lock:
                self.tts_speech_token_dict.pop(this_uuid, None)
                self.llm_end_dict.pop(this_uuid, None)
                self.hift_cache_dict.pop(this_uuid, None)
                self._llm_errors.pop(this_uuid, None)
            if getattr(self.flow, "supports_stream_cache", False):
                self.flow.drop_stream_cache(this_uuid)

    # ------------------------------------------------------------------ utterance-batched path (beyond the reference)
    @torch.no_grad()
    def tts_batch(self, texts: List[torch.Tensor], prompt_texts: List[torch.Tensor], llm_prompt_speech_tokens: List[torch.Tensor],
                  flow_prompt_speech_tokens: torch.Tensor, prompt_speech_feats: torch.Tensor, flow_embeddings: torch.Tensor,
                  forced: Optional[List[List[int]]] = None, to_host: bool = True):
        """B utterances (texts / prompts of any lengths) through LLM -> flow -> HiFT.  Returns wav (B,S) when all sequences
        generated the same number of tokens (teacher-forced), else a list of B waveforms."""
        toks = self.llm.generate_batch(texts, prompt_texts, llm_prompt_speech_tokens, forced=forced)
        wav = self._flow_hift(dict(flow_prompt_speech_tokens=flow_prompt_speech_tokens, prompt_speech_feats=prompt_speech_feats,
                                   flow_embeddings=flow_embeddings), toks)
        if isinstance(wav, list):   # sequences of different generated lengths
            return [w.cpu() for w in wav] if to_host else wav
        return wav.cpu() if to_host else wav.clone()   # never hand out a view of HiFT's per-shape workspace

    def cu_partition(self, llm_cu_slots: int, n_llm: int = 1):
        """(llm_streams, flow_stream): CU-masked streams giving the decode loop(s) ``llm_cu_slots`` of the 32 CUs of every XCD
        (one stream per concurrent decode loop, all on the same CUs) and flow + HiFT the rest (cached).  With both stages
        free to use all 256 CUs the decode step's short dependent kernels queue behind — and share CUs with — the flow
        GEMMs and run 1.6x slower while the two overlap (226 -> 368 ms per batch, tools/cumask_probe2.py); on disjoint CU
        sets neither disturbs the other."""
        from . import ops
        key = (llm_cu_slots, n_llm)
        if getattr(self, "_cu_partition_key", None) != key:
            self._release_partition()   # the previous split's streams are destroyed, not leaked
            k = llm_cu_slots
            self._cu_partition = ([ops.masked_stream(lambda s, x: s < k) for _ in range(n_llm)],
                                  ops.masked_stream(lambda s, x: s >= k),
                                  ops.masked_stream(lambda s, x: s >= k),   # a decode loop borrowing the idle flow CUs
                                  [ops.masked_stream(lambda s, x: s >= k) for _ in range(n_llm)],  # prefills, on the flow CUs
                                  ops.masked_stream(lambda s, x: s >= k))   # HiFT of batch i beside the flow of batch i + 1
            self._cu_partition_key = key
        return self._cu_partition

    def _release_partition(self):
        from . import ops
        part = getattr(self, "_cu_partition", None)
        if part is not None:
            torch.cuda.synchronize()
            flat = list(part[0]) + [part[1], part[2]] + list(part[3]) + list(part[4:5])
            for st in flat:
                ops.destroy_masked_stream(st)
        self._cu_partition, self._cu_partition_key = None, None

    def close(self):
        """Release what this model holds inside the HIP runtime (captured graphs of every stage / decode context, CU-masked
        streams) while the runtime is alive.  The model stays usable: graphs and streams are rebuilt on demand."""
        torch.cuda.synchronize()
        req_ctxs = [c for c, _ in (self._req_pool or [])]
        for ctx in list(getattr(self, "_llm_contexts", None) or [self.llm]) + req_ctxs:
            for g in getattr(ctx, "_graphs", {}).values():
                g.destroy()
            getattr(ctx, "_graphs", {}).clear()
        dec = getattr(self.flow, "decoder", None)
        for g in getattr(dec, "_graphs", {}).values():
            g.destroy()
        getattr(dec, "_graphs", {}).clear()
        self._release_partition()

    def llm_contexts(self, n: int):
        """``n`` decode contexts over the one set of LLM weights (Qwen2LM.new_context)."""
        ctxs = getattr(self, "_llm_contexts", None)
        if ctxs is None or ctxs[0] is not self.llm:
            ctxs = [self.llm]
        while len(ctxs) < n:
            ctxs.append(self.llm.new_context())
        self._llm_contexts = ctxs
        return ctxs[:n]

    @torch.no_grad()
    def warm_pipeline_shapes(self, b, n_tokens: int, llm_cu_slots: Optional[int] = None, llm_loops: Optional[int] = None):
        """Set-up, not work: run every shape the partitioned pipeline of ``tts_batches`` can meet once — each decode context at 1 ..
        ``llm_merge`` batches of rows (prefill workspace + the captured step graph of that row count), flow + HiFT at 1 .. ``flow_merge``
        batches on the flow CUs and on all CUs (the drain) — so that no graph capture or workspace allocation lands inside a run.
        ``b``: one batch dict as given to ``tts_batches`` (its ``forced`` lists, if any, are used for the two decode steps)."""
        k = self.llm_cu_slots if llm_cu_slots is None else llm_cu_slots
        n_llm = max(1, self.llm_loops if llm_loops is None else llm_loops)
        if not k:
            return
        llm_parts, flow_part, _, prefill_parts, hift_part = self.cu_partition(k, n_llm)
        B = len(b["texts"])
        forced = b.get("forced")
        for ctx, st, pf in zip(self.llm_contexts(n_llm), llm_parts, prefill_parts):
            for j in range(1, max(1, int(self.llm_merge)) + 1):
                if j * B > self.llm.max_batch:
                    break
                with torch.cuda.stream(st):
                    ctx.generate_batch(b["texts"] * j, b["prompt_texts"] * j, b["llm_prompt_speech_tokens"] * j,
                                       forced=None if forced is None else [list(t) for t in forced] * j, steps_per_poll=64, max_steps=2,
                                       prefill_stream=pf if self.prefill_on_flow_cus else None)
        est = getattr(getattr(self.flow, "decoder", None), "estimator", None)
        flow_full = torch.cuda.Stream(self.device)
        zero = torch.zeros(1, 1, 0)
        for budget, stream in (((32 - k) * 8, flow_part), (0, flow_full)):
            if est is not None:
                est.cu_budget = budget
            for j in range(1, max(1, int(self.flow_merge)) + 1):
                rep = lambda t: torch.cat([t] * j, dim=0)
                tok = torch.randint(0, self.flow.cfg.vocab_size, (j * B, n_tokens), dtype=torch.int32).to(self.device)
                with torch.cuda.stream(stream):
                    mel = self.flow.inference_batch(tok, rep(b["flow_prompt_speech_tokens"]), rep(b["prompt_speech_feats"]),
                                                    rep(b["flow_embeddings"])).contiguous().clone()
                stream.synchronize()
                with torch.cuda.stream(hift_part):
                    self.hift.inference(speech_feat=mel, cache_source=zero)
                hift_part.synchronize()
        if est is not None:
            est.cu_budget = 0
        torch.cuda.synchronize()

    def tts_batches(self, batches, to_host: bool = True, llm_cu_slots: Optional[int] = None, llm_loops: Optional[int] = None):
        """Generator over a list of utterance batches (each a dict of tts_batch's arguments), software-pipelined the way
        the reference overlaps its LLM thread with flow/HiFT (cli/model.py:62,119,189): the LLM decode of later batches
        (latency-bound) runs while flow + HiFT of batch i (throughput-bound GEMMs) execute.  A batch may carry ``on_start``
        (callable, e.g. the RCCL conditioning broadcast; called from the calling thread, in batch order) and ``on_done`` (callable,
        called from the calling thread once the batch's result has been collected, i.e. when no job reads its inputs any more:
        where a pipelined caller recycles the batch's conditioning buffer, cosyvoice_amd.dist.ConditioningRing).  ``batches`` may be
        a generator: a batch is taken from it (and may acquire its buffers) only when the pipeline admits it.  On the partitioned path the generator and
        ``on_start`` run with a private non-blocking stream current (ordered behind everything the caller enqueued before the call): GPU work they issue
        through torch is ordered before the batch's jobs without touching the caller's stream.

        ``llm_cu_slots`` (default ``self.llm_cu_slots``) > 0 partitions the GPU: the decode loops run on that many CUs
        of every XCD and flow + HiFT on the others, each from its own host thread, every captured graph replayed launch by
        launch (``ops.Graph.launch`` on a masked stream).  ``llm_loops`` (default ``self.llm_loops``) decode loops run
        concurrently on the decode CUs, each on its own batch with its own KV caches: a decode step is a chain of 124 short
        latency-bound kernels, and two interleaved chains deliver 1.5x the tokens per second of one.
        ``flow_merge`` > 1: the flow thread runs the oldest batch together with up to ``flow_merge`` - 1 batches right behind it whose tokens are
        already there as ONE flow + vocoder pass (same mel per utterance: the flow is batch-invariant; rows come back to their batches in order).
        ``llm_cu_slots=0`` keeps both stages on all CUs (two plain streams, one decode loop, no merging)."""
        k = self.llm_cu_slots if llm_cu_slots is None else llm_cu_slots
        n = self.llm_loops if llm_loops is None else llm_loops
        if k:
            yield from self._tts_batches_partitioned(batches, to_host, k, max(1, n))
        else:
            yield from self._tts_batches_shared(batches, to_host)

    def _flow_hift(self, b, toks):
        """Flow + HiFT of one batch.  Equal generated lengths (teacher-forced runs, the benchmark): one batched pass ->
        wav (B, S).  Free-running sampling ends every sequence at its own step: flow + HiFT then run per utterance (the
        flow runs once over the ragged batch, HiFT per utterance) -> list of B waveforms (S_b,)."""
        n = len(toks[0])
        zero = torch.zeros(1, 1, 0)
        if all(len(t) == n for t in toks):
            tok = torch.tensor(toks, dtype=torch.int32, device=self.device)
            mel = self.flow.inference_batch(tok, b["flow_prompt_speech_tokens"], b["prompt_speech_feats"], b["flow_embeddings"])
            wav, _ = self.hift.inference(speech_feat=mel.contiguous(), cache_source=zero)
            return wav
        # one batched flow pass over the ragged batch (rows padded to the longest, tails masked: exact), HiFT per utterance
        # (its symmetric convolutions would see the padding)
        B = len(toks)
        mels = self.flow.inference_ragged([torch.tensor(t, dtype=torch.int32) for t in toks],
                                          [b["flow_prompt_speech_tokens"][i] for i in range(B)],
                                          [b["prompt_speech_feats"][i] for i in range(B)], b["flow_embeddings"])
        outs = []
        for mel in mels:
            wav, _ = self.hift.inference(speech_feat=mel.unsqueeze(0).contiguous(), cache_source=zero)
            outs.append(wav[0].clone())
        return outs

    def _tts_batches_partitioned(self, batches, to_host, k, n_llm):
        import queue
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        llm_parts, flow_part, borrow_part, prefill_parts, hift_part = self.cu_partition(k, n_llm)
        est = getattr(getattr(self.flow, "decoder", None), "estimator", None)
        if est is not None:
            est.cu_budget = (32 - k) * 8   # the flow's launches run on that many CUs: its row-block kernels size their tiles for it
        flow_full = torch.cuda.Stream(self.device)   # all CUs: the last batch's flow + HiFT run after every decode loop ended
        caller = torch.cuda.current_stream()
        # Admission work (drawing a batch from ``batches``, its on_start, the `ready` event) runs on a private non-blocking stream, not on the
        # caller's: when that is the legacy default stream every operation on it — a conditioning copy, an event record — is an implicit barrier
        # against all the CU-masked streams of the pipeline (they are blocking streams), i.e. the whole pipeline drained once per admitted
        # batch (625 -> 638 audio-s/s at 20 passes, 639 -> 664 at 32, profiles/r03_flow_merge_sweeps.log).
        adm = torch.cuda.Stream(self.device)
        adm.wait_stream(caller)   # whatever the caller enqueued before the call is visible to the first batch
        ctxs = queue.Queue()
        for c, st, pf in zip(self.llm_contexts(n_llm), llm_parts, prefill_parts):
            ctxs.put((c, st, pf))

        def llm_job(bs, ready, borrow):
            """One decode job over the utterances of ``bs`` (``llm_merge`` consecutive batches share one token loop: the step
            is a weight stream, 16 rows cost 1.3x the time of 8) -> the token lists of each batch."""
            ctx, own, pf = ctxs.get()
            try:
                # first batch of the run: the flow CUs are idle, decode there (more CUs, nothing to disturb)
                stream = borrow_part if borrow else own
                cat = lambda key: [x for b in bs for x in b[key]]
                forced = cat("forced") if all(b.get("forced") is not None for b in bs) else None
                t_job = time.perf_counter()
                with torch.no_grad(), torch.cuda.stream(stream):
                    stream.wait_event(ready)   # inputs the caller produced / the conditioning broadcast
                    # the prefill is throughput-bound GEMM work: it runs beside flow + HiFT on their (three times larger)
                    # CU share, the decode partition only runs the token loop
                    toks = ctx.generate_batch(cat("texts"), cat("prompt_texts"), cat("llm_prompt_speech_tokens"),
                                              forced=forced, steps_per_poll=64,
                                              prefill_stream=None if (borrow or not self.prefill_on_flow_cus) else pf)
                out, o = [], 0
                for b in bs:
                    out.append(toks[o:o + len(b["texts"])])
                    o += len(b["texts"])
                if self.pipeline_stats is not None:   # generate_batch ends with a D2H of the token buffer: the loop is drained
                    self.pipeline_stats.append(("llm", len(bs), t_job, time.perf_counter()))
                return out
            finally:
                ctxs.put((ctx, own, pf))
                with llm_state_lock:
                    llm_state["running"] -= 1

        def split_rows(wav, counts, done):
            """result of a flow group -> one (wav, done) per batch of the group (rows in batch order)"""
            if isinstance(wav, list):   # ragged: per-utterance waveforms
                out, o = [], 0
                for c in counts:
                    out.append((wav[o:o + c], done))
                    o += c
                return out
            return [(w, done) for w in torch.split(wav, counts, dim=0)] if len(counts) > 1 else [(wav, done)]

        def hift_job(mel, mel_ready, t_job, counts):
            """HiFT (+ D2H) of one equal-length flow group on its own stream over the flow CUs: the vocoder of group i fills the CUs the
            flow of group i + 1 leaves idle (the tail round of every attention launch, tile quantisation of the row-block
            kernels) instead of extending the flow stream's critical path (17 of a 91 ms job on all CUs)."""
            with torch.no_grad(), torch.cuda.stream(hift_part):
                hift_part.wait_event(mel_ready)
                mel.record_stream(hift_part)
                wav, _ = self.hift.inference(speech_feat=mel, cache_source=torch.zeros(1, 1, 0))
                if to_host:
                    res = split_rows(wav.cpu(), counts, None)
                    if self.pipeline_stats is not None:
                        self.pipeline_stats.append(("flow", len(counts), t_job, time.perf_counter()))
                    return res
                wav = wav.clone()
                done = torch.cuda.Event()
                done.record(hift_part)
                return split_rows(wav, counts, done)

        keys = ("flow_prompt_speech_tokens", "prompt_speech_feats", "flow_embeddings")

        def flow_job(grp, srcs, readies, stream):
            """Flow + HiFT of ``grp`` = consecutive batches whose tokens are ready (``flow_merge`` of them at most, same prompt lengths) as a single
            pass: the row-block kernels run in whole rounds of workgroups over the stream's CUs, so 16 or 24 utterances cost less per utterance
            than 8 (tools/flow_share_probe.py: 13.9 / 12.3 / 11.5 ms per utterance at 8 / 16 / 24 on the 192-CU share).  ``srcs``: per batch its
            decode job's future and its index in that job.  -> one (wav, done) per batch."""
            per_b = [lf.result()[i] for lf, i in srcs]
            counts = [len(tb) for tb in per_b]
            toks = [t for tb in per_b for t in tb]
            ready_evs = []
            for r in readies:
                if all(r is not q for q in ready_evs):
                    ready_evs.append(r)
            b = grp[0] if len(grp) == 1 else None   # else concatenated on the job's stream, behind `ready` (below): never on this thread's default stream
            merged = lambda: {k_: torch.cat([bb[k_] for bb in grp], dim=0) for k_ in keys}
            t_job = time.perf_counter()
            with llm_state_lock:
                # every decode job of the run has ended and none is left to submit: the decode CUs are idle for good, so the remaining
                # flow + HiFT passes (the pipeline's drain: up to 2 x llm_merge batches) take all CUs, not only the last batch
                if llm_state["running"] == 0 and llm_state["all_submitted"]:
                    stream = flow_full
            if est is not None:   # tile sizes of the row-block kernels (and the solver graph that bakes them in) follow the CUs of the stream
                est.cu_budget = 0 if stream is flow_full else (32 - k) * 8
            n_t = len(toks[0])
            if self.overlap_hift and all(len(t) == n_t for t in toks):
                with torch.no_grad(), torch.cuda.stream(stream):
                    for r in ready_evs:
                        stream.wait_event(r)
                    stream.wait_stream(flow_part if stream is flow_full else flow_full)  # workspaces of the previous group
                    if b is None:
                        b = merged()
                    tok = torch.tensor(toks, dtype=torch.int32, device=self.device)
                    mel = self.flow.inference_batch(tok, b["flow_prompt_speech_tokens"], b["prompt_speech_feats"], b["flow_embeddings"])
                    mel = mel.contiguous().clone()   # the flow's output buffer is rewritten by the next group
                    mel_ready = torch.cuda.Event()
                    mel_ready.record(stream)
                return hift_pool.submit(hift_job, mel, mel_ready, t_job, counts)   # one worker: vocoder workspaces are used in group order
            with torch.no_grad(), torch.cuda.stream(stream):
                for r in ready_evs:
                    stream.wait_event(r)
                stream.wait_stream(flow_part if stream is flow_full else flow_full)  # workspaces of the previous group
                if self.overlap_hift:
                    hift_pool.submit(lambda: None).result()   # earlier groups' vocoder jobs are enqueued ...
                    stream.wait_stream(hift_part)             # ... and done with the vocoder workspaces before this one uses them
                if b is None:
                    b = merged()
                wav = self._flow_hift(b, toks)
                if to_host:
                    res = split_rows([w.cpu() for w in wav] if isinstance(wav, list) else wav.cpu(), counts, None)
                    if self.pipeline_stats is not None:
                        self.pipeline_stats.append(("flow", len(grp), t_job, time.perf_counter()))
                    return res
                # the equal-length path returns a view of HiFT's per-shape workspace, which the next flow_job overwrites on this
                # stream as soon as its tokens arrive: hand the consumer its own copy (made here, ordered before `done`)
                wav = [w.clone() for w in wav] if isinstance(wav, list) else wav.clone()
                done = torch.cuda.Event()
                done.record(stream)
                return split_rows(wav, counts, done)

        from concurrent.futures import Future
        flow_q, flow_cv, flow_state = deque(), threading.Condition(), {"closed": False, "abort": False}

        def same_layout(a, c):
            return all(a[k_].shape[1:] == c[k_].shape[1:] for k_ in keys)

        def flow_worker():
            """The flow thread: takes the oldest batch whose tokens it then waits for, adds the batches right behind it whose tokens are ALREADY
            there (up to ``flow_merge``, same prompt lengths) and runs them as one pass — groups form across decode jobs when the flow side has
            a backlog and never wait for tokens when it has none (the pipeline's fill)."""
            while True:
                with flow_cv:
                    while not flow_q and not flow_state["closed"]:
                        flow_cv.wait()
                    if not flow_q:
                        return
                    grp = [flow_q.popleft()]
                t0 = grp[0]
                try:
                    if flow_state["abort"]:
                        raise RuntimeError("tts_batches closed before this batch was vocoded")
                    t0["lf"].result()
                    with flow_cv:
                        while (len(grp) < fmerge and flow_q and flow_q[0]["lf"].done() and flow_q[0]["lf"].exception() is None
                               and same_layout(t0["b"], flow_q[0]["b"])):
                            grp.append(flow_q.popleft())
                    res = flow_job([t["b"] for t in grp], [(t["lf"], t["idx"]) for t in grp], [t["ready"] for t in grp], flow_part)
                except BaseException as e:
                    for t in grp:
                        t["fut"].set_exception(e)
                    continue

                def deliver(r, grp=grp):
                    try:
                        if hasattr(r, "result"):
                            r = r.result()
                        for t, x in zip(grp, r):
                            t["fut"].set_result(x)
                    except BaseException as e:
                        for t in grp:
                            if not t["fut"].done():
                                t["fut"].set_exception(e)
                if hasattr(res, "add_done_callback"):   # the equal-length overlap path hands back its vocoder job
                    res.add_done_callback(deliver)
                else:
                    deliver(res)

        def collect(fut):
            wav, done = fut.result()
            if done is not None:
                caller.wait_event(done)
            return wav

        it = iter(batches)

        def draw():
            with torch.cuda.stream(adm):
                return next(it, None)
        nxt = draw()
        inflight = deque()
        first = True
        merge = max(1, int(getattr(self, "llm_merge", 1)))
        fmerge = max(1, int(getattr(self, "flow_merge", 1)))
        llm_state_lock = threading.Lock()
        llm_state = {"running": 0, "all_submitted": False}
        # pipeline fill: the first decode jobs are small so that the flow stream gets its first batches early and is then fed without a
        # gap while the job size grows to `merge` (1, 2, 2, 3, merge, ...: with 4 batches per job from the start the flow CUs idled ~0.2 s)
        ramp = [int(v) for v in os.environ.get("CV_LLM_RAMP", ",".join(str(v) for v in self.llm_ramp)).split(",") if v]
        n_jobs = 0
        with ThreadPoolExecutor(max_workers=n_llm) as llm_pool, ThreadPoolExecutor(max_workers=1) as flow_pool, \
                ThreadPoolExecutor(max_workers=1) as hift_pool:
            flow_pool.submit(flow_worker)
            try:
                while nxt is not None or inflight:
                    # keep n_llm decode loops busy plus one job queued behind them.  A job is admitted only when ALL its batches fit: a job cut short
                    # by the in-flight cap (one batch in a 4-batch slot) occupies a decode loop for 70 % of a full job's time
                    job_batches = min(merge, ramp[n_jobs]) if n_jobs < len(ramp) else merge
                    while nxt is not None and (not inflight or len(inflight) + job_batches <= (n_llm + 2) * merge):
                        bs, rows = [], 0
                        while nxt is not None and len(bs) < job_batches and rows + len(nxt["texts"]) <= self.llm.max_batch:
                            rows += len(nxt["texts"])
                            bs.append(nxt)
                            nxt = draw()
                        if not bs:
                            raise ValueError(f"a batch of {len(nxt['texts'])} utterances exceeds the LLM's max_batch {self.llm.max_batch}")
                        ready = torch.cuda.Event()
                        with torch.cuda.stream(adm):
                            for b in bs:
                                if b.get("on_start") is not None:
                                    b["on_start"]()   # e.g. the conditioning broadcast: same order on every rank, never from worker threads
                            ready.record(adm)
                        n_jobs += 1
                        with llm_state_lock:
                            llm_state["running"] += 1
                            llm_state["all_submitted"] = nxt is None
                        lf = llm_pool.submit(llm_job, bs, ready, first)
                        with flow_cv:
                            for i, b in enumerate(bs):
                                t = dict(b=b, lf=lf, idx=i, ready=ready, fut=Future())
                                flow_q.append(t)
                                inflight.append((b, t["fut"]))
                            flow_cv.notify()
                        first = False
                        job_batches = min(merge, ramp[n_jobs]) if n_jobs < len(ramp) else merge
                    b_done, fut = inflight.popleft()
                    try:
                        wav = collect(fut)
                    finally:
                        if b_done.get("on_done") is not None:
                            b_done["on_done"]()   # every job of this batch (decode, flow, vocoder) has ended: its conditioning slot is free
                    yield wav
            finally:
                with flow_cv:   # normal end: the queue is empty; a consumer that stopped early: the flow thread fails what is left instead of running it
                    flow_state["closed"] = True
                    flow_state["abort"] = bool(flow_q)
                    flow_cv.notify()
        if est is not None:
            est.cu_budget = 0

    @torch.no_grad()
    def _tts_batches_shared(self, batches, to_host):
        llm_stream = self.llm_context.stream if hasattr(self.llm_context, "stream") else torch.cuda.Stream()
        pending, pending_b = None, None
        import itertools
        for b in itertools.chain(batches, [None]):   # lazily: a batch takes its buffers when its turn comes
            toks = None
            if b is not None:
                with torch.cuda.stream(llm_stream):
                    if b.get("on_start") is not None:
                        b["on_start"]()
                    ready = torch.cuda.Event()
                    ready.record(llm_stream)
                    toks = self.llm.generate_batch(b["texts"], b["prompt_texts"], b["llm_prompt_speech_tokens"], forced=b.get("forced"))
            if pending is not None:
                if isinstance(pending, list):
                    out = [w.cpu() for w in pending] if to_host else pending
                else:
                    out = pending.cpu() if to_host else pending.clone()
                if not to_host:
                    torch.cuda.current_stream().synchronize()   # on_done promises that nothing reads the batch's inputs any more
                if pending_b.get("on_done") is not None:
                    pending_b["on_done"]()
                yield out
                pending = None
            if toks is not None:
                torch.cuda.current_stream().wait_event(ready)  # conditioning of this batch is in place
                pending, pending_b = self._flow_hift(b, toks), b


class CosyVoiceModel(CosyVoice2Model):
    """Drop-in for the reference's ``CosyVoiceModel`` (/root/reference/cosyvoice/cli/model.py:27-292) — the v1 wiring, which
    the fork also drives CosyVoice2 modules with (bin/inference_online_codec.py): the estimator attends over the whole
    sequence (``static_chunk_size = 0``, :49-50; the encoder's chunk size is left alone), streaming works on hop-sized
    token windows with ``token_overlap_len`` = 20 tokens of overlap (every chunk sees only the prompt and its own window, the
    consumed tokens are dropped), consecutive mels are cross-faded over ``mel_overlap_len`` frames and the vocoder keeps a
    20-frame mel / source / speech cache (:52-60, :130-172).  ``vc`` feeds source speech tokens instead of running the LLM."""

    def __init__(self, llm, flow, hift, fp16: bool = False, sr: int = 22050):
        super().__init__(llm, flow, hift, fp16)
        self.sample_rate = sr
        self.flow.encoder.static_chunk_size = 0              # CosyVoice2Model.__init__ set 2 * frame rate; v1 never does
        self.flow.decoder.estimator.static_chunk_size = 0    # model.py:49-50
        self.token_min_hop_len = 2 * self.flow.input_frame_rate
        self.token_max_hop_len = 4 * self.flow.input_frame_rate
        self.token_overlap_len = 20
        self.mel_overlap_len = int(self.token_overlap_len / self.flow.input_frame_rate * sr / 256)
        self.mel_window = np.hamming(2 * self.mel_overlap_len)
        self.mel_cache_len = 20
        self.source_cache_len = int(self.mel_cache_len * 256)
        self.speech_window = np.hamming(2 * self.source_cache_len)
        self.mel_overlap_dict = {}
        self.flow_cache_dict = {}

    def _token2wav(self, token, prompt_token=torch.zeros(1, 0, dtype=torch.int32), prompt_feat=torch.zeros(1, 0, 80),
                  embedding=torch.zeros(0, 512), uuid_="0", finalize=False, speed=1.0):
        # model.py:130-172
        n = lambda t: torch.tensor([t.shape[1]], dtype=torch.int32)
        mel, self.flow_cache_dict[uuid_] = self.flow.inference(
            token=token.to(self.device), token_len=n(token), prompt_token=prompt_token.to(self.device), prompt_token_len=n(prompt_token),
            prompt_feat=prompt_feat.to(self.device), prompt_feat_len=n(prompt_feat), embedding=embedding.to(self.device),
            flow_cache=self.flow_cache_dict[uuid_], sample_rate=self.sample_rate)
        if self.mel_overlap_dict[uuid_].shape[2] != 0:
            mel = fade_in_out(mel, self.mel_overlap_dict[uuid_], self.mel_window)
        cache = self.hift_cache_dict[uuid_]
        cache_source = torch.zeros(1, 1, 0)
        if cache is not None:
            mel = torch.concat([cache["mel"], mel], dim=2)
            cache_source = cache["source"]
        if not finalize:
            self.mel_overlap_dict[uuid_] = mel[:, :, -self.mel_overlap_len:].clone()
            mel = mel[:, :, :-self.mel_overlap_len]
            speech, source = self.hift.inference(speech_feat=mel, cache_source=cache_source)
            speech, source = speech.clone(), source.clone()
            if cache is not None:
                speech = fade_in_out(speech, cache["speech"], self.speech_window)
            self.hift_cache_dict[uuid_] = {"mel": mel[:, :, -self.mel_cache_len:].clone(),
                                           "source": source[:, :, -self.source_cache_len:].clone(),
                                           "speech": speech[:, -self.source_cache_len:].clone()}
            return speech[:, :-self.source_cache_len]
        if speed != 1.0:
            assert cache is None, "speed change only support non-stream inference mode"
            mel = F.interpolate(mel, size=int(mel.shape[2] / speed), mode="linear")
        speech, _ = self.hift.inference(speech_feat=mel, cache_source=cache_source)
        speech = speech.clone()
        if cache is not None:
            speech = fade_in_out(speech, cache["speech"], self.speech_window)
        return speech

    def _open(self, uuid_, tokens, ended):
        with self.lock:
            self.tts_speech_token_dict[uuid_], self.llm_end_dict[uuid_] = tokens, ended
            self.hift_cache_dict[uuid_] = None
            self.mel_overlap_dict[uuid_] = torch.zeros(1, 80, 0)
            self.flow_cache_dict[uuid_] = torch.zeros(1, 80, 0, 2)

    def _close(self, uuid_):
        with self.lock:
            for d in (self.tts_speech_token_dict, self.llm_end_dict, self.mel_overlap_dict, self.hift_cache_dict, self.flow_cache_dict, self._llm_errors):
                d.pop(uuid_, None)

    def _emit(self, uuid_, producer, flow_prompt_speech_token, prompt_speech_feat, flow_embedding, stream, speed, poll):
        """The token -> waveform schedule shared by tts (:173-238) and vc (:240-292): hop + overlap windows while streaming,
        one pass over everything otherwise."""
        kw = dict(prompt_token=flow_prompt_speech_token, prompt_feat=prompt_speech_feat, embedding=flow_embedding, uuid_=uuid_)
        toks = lambda k=None: torch.tensor(self.tts_speech_token_dict[uuid_][:k]).unsqueeze(dim=0)
        if stream:
            hop = self.token_min_hop_len
            while True:
                if poll:
                    time.sleep(poll)
                if len(self.tts_speech_token_dict[uuid_]) >= hop + self.token_overlap_len:
                    yield {"tts_speech": self.token2wav(token=toks(hop + self.token_overlap_len), finalize=False, **kw).cpu()}
                    with self.lock:
                        self.tts_speech_token_dict[uuid_] = self.tts_speech_token_dict[uuid_][hop:]
                    hop = min(self.token_max_hop_len, int(hop * self.stream_scale_factor))
                if self.llm_end_dict[uuid_] is True and len(self.tts_speech_token_dict[uuid_]) < hop + self.token_overlap_len:
                    break
            if producer is not None:
                producer.join()
                self._raise_llm_error(uuid_)
            yield {"tts_speech": self.token2wav(token=toks(), finalize=True, **kw).cpu()}
        else:
            if producer is not None:
                producer.join()
                self._raise_llm_error(uuid_)
            yield {"tts_speech": self.token2wav(token=toks(), finalize=True, speed=speed, **kw).cpu()}

    def _raise_llm_error(self, uuid_):
        err = self._llm_errors.pop(uuid_, None)
        if err is not None:
            raise err

    def tts(self, text, flow_embedding, llm_embedding=torch.zeros(0, 512), prompt_text=torch.zeros(1, 0, dtype=torch.int32),
            llm_prompt_speech_token=torch.zeros(1, 0, dtype=torch.int32), flow_prompt_speech_token=torch.zeros(1, 0, dtype=torch.int32),
            prompt_speech_feat=torch.zeros(1, 0, 80), stream=False, speed=1.0, **kwargs):
        this_uuid = str(uuid.uuid1())
        self._open(this_uuid, [], False)
        p = threading.Thread(target=self.llm_job, args=(text, prompt_text, llm_prompt_speech_token, llm_embedding, this_uuid))
        p.start()
        try:
            yield from self._emit(this_uuid, p, flow_prompt_speech_token, prompt_speech_feat, flow_embedding, stream, speed, 0.1)
        finally:
            p.join()
            self._close(this_uuid)

    def vc(self, source_speech_token, flow_prompt_speech_token, prompt_speech_feat, flow_embedding, stream=False, speed=1.0, **kwargs):
        this_uuid = str(uuid.uuid1())
        self._open(this_uuid, source_speech_token.flatten().tolist(), True)
        try:
            yield from self._emit(this_uuid, None, flow_prompt_speech_token, prompt_speech_feat, flow_embedding, stream, speed, 0.0)
        finally:
            self._close(this_uuid)
