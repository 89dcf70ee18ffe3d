#!/usr/bin/env python3
"""bench.py — audio-seconds per wall-second of the LLM -> flow -> HiFT hot path on MI355X.

Workload (BASELINE.json configs[3], "Full LLM->flow->hifigan pipeline, batch 8 x 10 s utterances, 1xMI355X with
hipGraph token loop"): per GPU 8 utterances, each a 10 s prompt (N_p=250 prompt speech tokens,
T_p=500 prompt mel frames, 10 prompt-text + 20 text ids -> prefill length 282) generating N_g=250 teacher-forced
speech tokens = 10.0 s of 24 kHz audio; flow T = 1000 frames, 10 CFG Euler steps; HiFT 500 frames -> 240 000 samples.
One "step" = the whole batch through the pipeline (prefill + 249 decode steps incl. on-device sampling, flow encoder +
solver, HiFT, waveform D2H).  The K timed steps are software-pipelined (LLM decode of later steps overlaps flow + HiFT of
step i, as the reference overlaps its LLM thread with flow/HiFT); all K steps' work, fill and drain included, lies inside
the timed region; up to `--flow-merge` = 3 consecutive passes whose tokens are ready share ONE flow + HiFT launch sequence (same mel per utterance:
the flow is batch-invariant, tests/test_model_gpu.py), and every graph capture / workspace allocation happens in set-up
(`CosyVoice2Model.warm_pipeline_shapes`).  Synthetic inputs + key-seeded random weights of the reference's architecture (no checkpoint exists
offline); inputs are resident in HBM before the timed region.

N>1 (BASELINE configs[4]): one process per GPU (torchrun); a GLOBAL list of 8 x N utterances is sharded round-robin
(cosyvoice_amd.dist.shard_utterances: weak scaling, 8 per rank), ONE broadcast per step of the shared prompt conditioning
from rank 0 (dist.pack / broadcast / unpack_conditioning; RCCL over xGMI), no other collective on the data
path; barrier + synchronize bracket the timed region, MAX over ranks (dist.max_over_ranks).  tests/test_dist_cpu.py drives
exactly those functions with 2 gloo ranks.

Besides the headline the line carries: per-stage times of one batch on all CUs (`stages_ms`), the other single-GPU BASELINE
configs (`c2_flow_only`, `c3_hift_only` with HBM- and MFMA-roofline fractions, exact-f32 beside bf16x3 products), three
roofline blocks (`roofline` = the by-time dominant kernel of the pipeline, the flow's flash attention; `roofline_flow` = the
flow stage on SURVEY §8d's 4.937 TF per utterance; `roofline_decode` = the decode step at step level, 727.6 MB per step, and
its largest weight stream), and `cpu_baseline` (oracle, median of 3 warmed samples per component).
"""
import argparse
import json
import os
import statistics
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between the ranks of one node needs it on this driver

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

UTT_PER_GPU = 8
ATTN_PMC = "r03_tblock_hbm.json"            # profiles/: tools/pmc_hbm.py over tools/tblock_bench.py 16 1000
DECODE_PMC = "r03_roofline_pmc_split_b%d.json"   # profiles/: tools/pmc_hbm.py over tools/roofline_pmc.py <rows> split
HIFT_PMC = "r03_hift_hbm.json"              # profiles/: tools/pmc_hbm.py over tools/hift_pmc.py
N_PROMPT, N_GEN, L_TEXT, L_PTEXT = 250, 250, 20, 10
AUDIO_S_PER_UTT = N_GEN * 2 * 480 / 24000.0  # 10.0 s


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def make_inputs(lc, fc, seed, n_utts=UTT_PER_GPU):
    """The global utterance list (texts + forced token lists) and the one shared speaker prompt, identical on every rank."""
    g = torch.Generator().manual_seed(seed)
    texts = [torch.randint(0, lc.vocab_size, (1, L_TEXT), generator=g, dtype=torch.int32) for _ in range(n_utts)]
    forced = [torch.randint(0, lc.speech_token_size, (N_GEN,), generator=g).tolist() for _ in range(n_utts)]
    # shared prompt conditioning (one speaker prompt for the whole batch: what the broadcast carries)
    ptext = torch.randint(0, lc.vocab_size, (1, L_PTEXT), generator=g, dtype=torch.int32)
    pspeech = torch.randint(0, lc.speech_token_size, (1, N_PROMPT), generator=g, dtype=torch.int32)
    pfeat = torch.clamp(torch.randn(1, 2 * N_PROMPT, 80, generator=g) * 2 - 6, -11.5, 2.0)
    emb = torch.randn(1, fc.spk_embed_dim, generator=g)
    return texts, forced, ptext, pspeech, pfeat, emb


def cpu_model_string():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(lsd, fsd, hsd, lc, fc, hc):
    """The oracle (CPU port of the reference arithmetic) timed on this host's cores on a bounded sample of the SAME workload:
    every component is run once to warm (thread pool, allocator, page faults) and then timed (median of 3 for C4, of 2 for C1), and the
    per-utterance time is EXTRAPOLATED linearly in the repeated units (decode steps, Euler steps, mel frames) — `kind` says so.
    Two workloads: the headline's (C4 utterance: 10 s prompt, prefill 282, flow T = 1000) and BASELINE config C1 (3 s prompt + 20-token
    text: prefill 107, flow T = 650), the reference's own CPU-runnable case."""
    from oracle import flow as of
    from oracle import hift as oh
    from oracle import llm as ol
    # cores: the GPU box leases one GPU together with a 16-core share of the host CPU (gpurun's process guard sizes worker pools to
    # it; the driver runs 8 ranks per node, 8 x 16 = the host's 128 hardware threads), so 16 threads is "all the cores this GPU's
    # process may use".  torch's default (every hardware thread of the host) also runs these small ops slower.
    cores = min(16, torch.get_num_threads())
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(1)

    def med(fn, n=3):
        fn()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts)

    def sample(n_prompt, reps):
        L = 1 + L_PTEXT + L_TEXT + 1 + n_prompt
        x = torch.randn(1, L, lc.hidden_size, generator=g) * 0.02
        x1 = torch.randn(1, 1, lc.hidden_size, generator=g) * 0.02

        def prefill():
            cache = ol.KVCache(lc.num_layers)
            y = ol.qwen2_forward(lsd, lc, x, cache)
            ol.logits_to_logp(lsd, y[:, -1])
            return cache
        t_pre = med(prefill, reps)
        cache = prefill()

        def step4():   # 4 decode steps from context L on (the cache grows by 4 per call)
            for _ in range(4):
                y = ol.qwen2_forward(lsd, lc, x1, cache)
                ol.logits_to_logp(lsd, y[:, -1])
        t_step = med(step4, reps) / 4
        N = n_prompt + N_GEN
        xs = torch.randn(1, N, fc.input_size, generator=g)
        t_enc = med(lambda: of.encoder_forward(fsd, fc, xs, torch.tensor([N]), 0), reps)
        T = 2 * N
        a = lambda *s: torch.randn(*s, generator=g)
        ex = (a(2, 80, T), torch.ones(2, 1, T), a(2, 80, T), torch.tensor([0.5, 0.5]), a(2, 80), a(2, 80, T))
        t_est = med(lambda: of.estimator_forward(fsd, fc, *ex), reps)
        return L, N, T, t_pre, t_step, t_enc, t_est

    with torch.inference_mode():
        L, N, T, t_pre, t_step, t_enc, t_est = sample(N_PROMPT, 3)
        mel = torch.clamp(torch.randn(1, 80, 100, generator=g) * 2 - 6, -11.5, 2.0)
        ph, nz = oh.draw_source_randoms(hc, 1, 100 * hc.total_upsample, seed=2)
        t_hift = med(lambda: oh.inference(hsd, hc, mel, None, ph, nz)) * 5
        c1 = sample(75, 2)
    t_dec = (N_GEN - 1) * t_step
    per_utt = t_pre + t_dec + t_enc + fc.n_timesteps * t_est + t_hift
    L1, N1, T1, p1, s1, e1, est1 = c1
    per_utt_c1 = p1 + (N_GEN - 1) * s1 + e1 + fc.n_timesteps * est1 + t_hift
    return {"value": round(AUDIO_S_PER_UTT / per_utt, 4), "unit": "audio-seconds/sec", "cores": cores, "kind": "port, extrapolated sample",
            "cores_note": "16 = the CPU share the GPU box leases with one GPU (8 ranks x 16 threads = the host's 128 hardware threads)",
            "cpu_model": cpu_model_string(),
            "stages_s": {"llm_prefill": round(t_pre, 3), "llm_decode": round(t_dec, 3), "flow_encoder": round(t_enc, 3),
                         "flow_solver": round(fc.n_timesteps * t_est, 3), "hift": round(t_hift, 3), "per_utterance": round(per_utt, 2)},
            "sample": (f"oracle fp32, 1 utterance of the same workload, every component warmed once then median of 3: prefill L={L} "
                       f"({t_pre:.2f}s) + 4 decode steps ({t_step*1e3:.1f} ms/step, x249) + flow encoder N={N} ({t_enc:.2f}s) + 1 estimator "
                       f"call T={T} ({t_est:.2f}s, x10) + HiFT 100/500 frames (x5 = {t_hift:.2f}s)"),
            "c1": {"workload": "BASELINE C1: zero-shot, 3 s prompt (75 tokens, 150 mel frames) + 20-token text -> 250 tokens = 10 s of audio; CPU fp32",
                   "value": round(AUDIO_S_PER_UTT / per_utt_c1, 4), "unit": "audio-seconds/sec", "rtf": round(per_utt_c1 / AUDIO_S_PER_UTT, 3),
                   "stages_s": {"llm_prefill": round(p1, 3), "llm_decode": round((N_GEN - 1) * s1, 3), "flow_encoder": round(e1, 3),
                                "flow_solver": round(fc.n_timesteps * est1, 3), "hift": round(t_hift, 3), "per_utterance": round(per_utt_c1, 2)},
                   "sample": (f"same method, median of 2: prefill L={L1} ({p1:.2f}s) + 4 decode steps ({s1*1e3:.1f} ms/step, x249) + flow encoder "
                              f"N={N1} ({e1:.2f}s) + 1 estimator call T={T1} ({est1:.2f}s, x10) + the HiFT sample above")}}


def ev_time(fn, n, warm=2):
    """Seconds per call: n calls between HIP events on torch's current stream, after `warm` untimed calls."""
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)   # 32 pipelined passes (~5 s): fill + drain (~0.5 s) stay inside the timed region
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the stage / C2 / C3 measurements (tuning runs)")
    ap.add_argument("--flow-dtype", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--llm-dtype", default="fp16", choices=["fp16", "bf16"],
                    help="16-bit operand type of the LLM (fp32 accumulate).  fp16 = what the reference's fp16 flag gives (llm.half(), cli/model.py:43-45) and "
                         "the one that meets the stated logit tolerance (L-inf 1.3e-2 <= 5e-2 vs the reference loop; bf16: 1.1e-1)")
    ap.add_argument("--llm-cu-slots", type=int, default=8,
                    help="CU slots per XCD (of 32) owned by the decode loops while they overlap flow+HiFT; 0 = no partition")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; gloo only to rehearse the multi-process path with several ranks on ONE GPU")
    ap.add_argument("--llm-merge", type=int, default=4,
                    help="consecutive 8-utterance batches decoded by one token loop (4: 32 rows = two 16-row MFMA groups share every weight stream)")
    ap.add_argument("--flow-merge", type=int, default=3,
                    help="consecutive batches of one decode job that share one flow + HiFT pass (24 utterances fill the flow's 192 CUs in whole "
                         "rounds of workgroups: 11.5 vs 13.9 ms per utterance at 8, tools/flow_share_probe.py)")
    ap.add_argument("--llm-loops", type=int, default=2,
                    help="concurrent decode loops (one utterance batch each, own KV caches) on the decode CUs")
    args = ap.parse_args()

    torch.set_num_threads(min(16, torch.get_num_threads()))  # CPU share per GPU on the box; keeps 8 ranks from oversubscribing
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    local_rank %= torch.cuda.device_count()   # one rank per GPU; the modulo only matters for the gloo rehearsal on one GPU
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    from cosyvoice_amd import dist as cd
    from cosyvoice_amd import ops
    from cosyvoice_amd.config import FlowConfig, HiftConfig, LlmConfig
    from cosyvoice_amd.flow import CausalMaskedDiffWithXvec
    from cosyvoice_amd.hift import HiFTGenerator
    from cosyvoice_amd.llm import Qwen2LM
    from cosyvoice_amd.model import CosyVoice2Model
    from cosyvoice_amd.weights import flow_state_dict, hift_state_dict, llm_state_dict

    lc, fc, hc = LlmConfig.full(), FlowConfig.full(), HiftConfig.v2()
    t0 = time.time()
    lsd, fsd, hsd = llm_state_dict(lc), flow_state_dict(fc), hift_state_dict(hc)
    log(f"[rank {rank}] synthetic weights generated in {time.time()-t0:.1f}s")
    fdt = torch.float16 if args.flow_dtype == "fp16" else torch.bfloat16
    ldt = torch.float16 if args.llm_dtype == "fp16" else torch.bfloat16
    llm = Qwen2LM(lc, dtype=ldt, max_batch=UTT_PER_GPU * max(1, args.llm_merge), ctx_max=704, max_out=N_GEN + 8)
    flow = CausalMaskedDiffWithXvec(fc, dtype=fdt)
    hift = HiFTGenerator(hc, dtype=torch.float32)
    model = CosyVoice2Model(llm, flow, hift, fp16=False).load_state_dicts(lsd, fsd, hsd)
    model.llm_merge = max(1, args.llm_merge)
    model.flow_merge = max(1, args.flow_merge)
    if model.flow_merge >= 3:
        model.llm_ramp = (2, 3, 3)   # merged flow passes want their first groups sooner and larger (profiles/r03_flow_merge_sweeps.log)
    # the fork drives CosyVoice2 modules through CosyVoiceModel wiring (full attention, model.py:50); the benchmark keeps
    # CosyVoice2Model's own chunk-50 encoder mask (model.py:314)
    flow.decoder.use_graph = True
    log(f"[rank {rank}] models on device in {time.time()-t0:.1f}s")

    # ---- the global utterance list (identical on every rank), sharded: rank r takes utterances {i : i mod world = r}
    n_global = UTT_PER_GPU * world
    texts, forced, ptext, pspeech, pfeat, emb = make_inputs(lc, fc, seed=100, n_utts=n_global)
    mine = cd.shard_utterances(n_global, world, rank)
    dev = torch.device("cuda", local_rank)
    B = len(mine)
    texts_d = [texts[i].to(dev) for i in mine]
    forced_m = [forced[i] for i in mine]
    # conditioning payload [prompt mel | speaker embedding | prompt speech tokens | prompt text ids], one flat fp32 buffer PER BATCH:
    # every pipeline pass has its own speaker prompt (a perturbed prompt mel / embedding, its own prompt speech tokens), so a payload
    # that reached a consumer late, early or overwritten would change that pass's audio.  Rank 0 holds all of them in HBM before the
    # timed region; the other ranks only ever see what the per-batch broadcast delivers.
    n_payloads = args.warmup + args.steps + 8
    _, layout = cd.pack_conditioning(pfeat, emb, pspeech, ptext)

    def payload_of(i):
        g = torch.Generator().manual_seed(7000 + i)
        pf = torch.clamp(pfeat + 0.05 * torch.randn(pfeat.shape, generator=g), -11.5, 2.0)
        em = emb + 0.1 * torch.randn(emb.shape, generator=g)
        ps = torch.randint(0, lc.speech_token_size, pspeech.shape, generator=g, dtype=torch.int32)
        return cd.pack_conditioning(pf, em, ps, ptext)[0]
    payloads = [payload_of(i).to(dev) for i in range(n_payloads)] if rank == 0 else None
    # one slot per batch in flight (cosyvoice_amd.dist.ConditioningRing): held from the batch's broadcast until tts_batches reports
    # the batch done, so a later broadcast never rewrites conditioning a running decode / flow job still reads
    merge = max(1, args.llm_merge)
    ring = cd.ConditioningRing((max(1, args.llm_loops) + 2) * merge + merge + 2, layout, dev)
    seq = {"i": 0}

    def make_batch(local=False):
        """``local``: rank 0 fills the slot itself, no collective (the single-rank measurements after the timed region)."""
        i = seq["i"] % n_payloads
        seq["i"] += 1
        slot = ring.acquire()
        buf = ring.slots[slot]

        def bcast():
            if rank == 0:
                buf.copy_(payloads[i])
            cd.broadcast_conditioning(buf, dist, src=0)
            ring.after_broadcast(slot)   # token ids -> the slot's int32 tensors, stream-ordered behind the broadcast
        pf, em, ps, pt = ring.tensors(slot)
        if local:
            buf.copy_(payloads[i])
            ring.after_broadcast(slot)
        return dict(texts=texts_d, prompt_texts=[pt] * B, llm_prompt_speech_tokens=[ps] * B,
                    flow_prompt_speech_tokens=ps.expand(B, -1), prompt_speech_feats=pf.expand(B, -1, -1),
                    flow_embeddings=em.expand(B, -1), forced=forced_m, on_start=bcast, on_done=lambda: ring.release(slot))

    def run_steps(n):
        """n pipeline passes ("steps", one batch of 8 utterances each): the decode loops of later passes (two at a time, on
        their CU share) overlap flow + HiFT of pass i (on the other CUs); every waveform is copied to the host inside the
        region.  Batches are made lazily: a pass takes its conditioning slot when the pipeline admits it."""
        last = None
        for wav in model.tts_batches((make_batch() for _ in range(n)), to_host=True, llm_cu_slots=args.llm_cu_slots, llm_loops=args.llm_loops):
            last = wav
        assert ring.in_use() == 0
        return last

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # set-up (untimed, like loading the weights): every graph capture / workspace allocation of the shapes the pipeline can meet — decode
    # contexts at 8 .. 8 x llm_merge rows, flow + HiFT groups of 1 .. flow_merge batches on the flow CUs and on all CUs
    wb = make_batch(local=(rank == 0))   # shapes only: the other ranks' slot holds zeros, nothing is broadcast
    model.warm_pipeline_shapes(wb, N_GEN, llm_cu_slots=args.llm_cu_slots, llm_loops=args.llm_loops)
    wb["on_done"]()
    if args.warmup > 0:
        w = run_steps(args.warmup)
        log(f"[rank {rank}] warmup: wav {tuple(w.shape)} absmax {w.abs().max().item():.3f}")
    fence()
    model.pipeline_stats = [] if os.environ.get("CV_PIPELINE_STATS") else None
    t0 = time.perf_counter()
    run_steps(args.steps)
    fence()
    elapsed = cd.max_over_ranks(time.perf_counter() - t0, dev, dist)
    if model.pipeline_stats:
        for kind in ("llm", "flow"):
            js = [(e - s_) * 1e3 / n for k, n, s_, e in model.pipeline_stats if k == kind]
            log(f"[rank {rank}] pipeline {kind} jobs: " + " ".join(f"{j:.0f}" for j in js) + " ms per batch")
        tmin = min(s_ for _, _, s_, _ in model.pipeline_stats)
        log("timeline: " + " | ".join(f"{k}{n} {1e3*(s_-tmin):.0f}-{1e3*(e-tmin):.0f}" for k, n, s_, e in sorted(model.pipeline_stats, key=lambda x: x[2])))
    model.pipeline_stats = None

    # what the collective layer saw: world size / backend of the process group and a SUM over ranks of the utterance counts
    dist_info = {"world_size": 1, "backend": None, "utterances_per_rank": [B], "utterances_total_allreduce": B}
    if dist is not None:
        cnt = torch.tensor([B], device=dev if args.backend == "nccl" else "cpu", dtype=torch.int64)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        dist_info = {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                     "utterances_per_rank": [len(cd.shard_utterances(n_global, world, r)) for r in range(world)],
                     "utterances_total_allreduce": int(cnt.item())}

    out = None
    if rank == 0:
        one = make_batch(local=True)
        # ---- batch-1 latency of the same pipeline (BASELINE metric: "batch 1 and 8"; north_star target RTF < 0.05)
        def run1():
            return model.tts_batch(one["texts"][:1], one["prompt_texts"][:1], one["llm_prompt_speech_tokens"][:1],
                                   one["flow_prompt_speech_tokens"][:1], one["prompt_speech_feats"][:1], one["flow_embeddings"][:1],
                                   forced=forced_m[:1], to_host=True)
        for _ in range(2):
            run1()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n1 = 3
        for _ in range(n1):
            run1()
        torch.cuda.synchronize()
        lat = (time.perf_counter() - t1) / n1
        b1 = {"latency_ms": round(lat * 1e3, 2), "rtf": round(lat / AUDIO_S_PER_UTT, 5), "audio_s_per_s": round(AUDIO_S_PER_UTT / lat, 2)}

        extras = {}
        if not args.no_extras:
            extras = measure_extras(model, llm, flow, hift, lc, fc, hc, one, forced_m, fdt, args)
        replay = ("each decode step is ONE captured hipGraph, replayed node by node (hipLaunchKernel per captured kernel) on CU-masked "
                  "streams because hipGraph replays ignore a stream's CU mask" if args.llm_cu_slots else
                  "each decode step is ONE captured hipGraph replayed with hipGraphLaunch")
        audio_s = AUDIO_S_PER_UTT * B * world * args.steps
        out = {
            "metric": "audio-seconds/sec (RTF^-1) end-to-end, 10s prompt, batch 8",
            "value": round(audio_s / elapsed, 3), "unit": "audio-seconds/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": f"llm {args.llm_dtype} / flow {args.flow_dtype} MFMA operands, fp32 accumulate; hift fp32 tensors, bf16x3 split-product MFMA",
            "data": "synthetic (key-seeded random weights of the reference architecture, teacher-forced 250 tokens)",
            "config": {"workload": "C4 full LLM->flow->HiFT pipeline, 8 utterances x 10 s per GPU, 10 s prompt "
                                   "(prefill 282, N_g 250, flow T 1000 x 10 CFG Euler steps, HiFT 500 frames); token loop: " + replay,
                       "utterances_per_gpu": B, "rtf": round(elapsed / audio_s, 6), "parallelism": f"utterance-parallel x{world}",
                       "llm_cu_slots_per_xcd": args.llm_cu_slots, "llm_decode_loops": args.llm_loops, "batches_per_decode_loop": args.llm_merge, "batches_per_flow_pass": args.flow_merge,
                       "flow_fused_tblock": bool(flow.decoder.estimator.fused), "batch1": b1, "dist": dist_info},
        }
        out.update(extras)
        if "roofline" not in out:
            out["roofline"] = measure_decode_roofline(llm, lc, rows=min(llm.max_batch, UTT_PER_GPU * max(1, args.llm_merge)))
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(lsd, fsd, hsd, lc, fc, hc)
        print(json.dumps(out), flush=True)
    fence()
    model.close()
    ops.close()
    if dist is not None:
        dist.destroy_process_group()


def measure_extras(model, llm, flow, hift, lc, fc, hc, one, forced_m, fdt, args):
    """Stage times of one batch (all CUs, stages run back to back, not overlapped), the flow / decode rooflines and BASELINE
    configs C2 / C3.  Rank 0, outside the timed region."""
    from cosyvoice_amd import ops
    from cosyvoice_amd.config import FlowConfig, HiftConfig
    from cosyvoice_amd.hift import HiFTGenerator
    from cosyvoice_amd.weights import hift_state_dict
    dev = "cuda"
    B = len(one["texts"])
    res = {}
    # ---------------- LLM: prefill (+ first token) vs the 249 decode steps, batch 8, graph replays on all CUs
    gb = lambda steps: llm.generate_batch(one["texts"], one["prompt_texts"], one["llm_prompt_speech_tokens"], forced=forced_m,
                                          steps_per_poll=64, max_steps=steps)
    t_pre = ev_time(lambda: gb(1), 3, warm=1)
    t_llm = ev_time(lambda: gb(None), 2, warm=1)
    toks = gb(None)
    # ---------------- flow: encoder vs solver (+ glue), batch 8, T = 1000
    tok = torch.tensor(toks, dtype=torch.int32, device=dev)
    fargs = (tok, one["flow_prompt_speech_tokens"], one["prompt_speech_feats"], one["flow_embeddings"])
    t_flow = ev_time(lambda: flow.inference_batch(*fargs), 4, warm=2)
    N = N_PROMPT + N_GEN
    ews = flow.encoder._workspace(B, N)
    t_enc = ev_time(lambda: flow.encoder.forward_tokens(ews["tok"], B, N), 4, warm=1)
    mel = flow.inference_batch(*fargs).clone().contiguous()
    zero = torch.zeros(1, 1, 0)
    t_hift = ev_time(lambda: hift.inference(speech_feat=mel, cache_source=zero), 6, warm=2)
    hift_x = HiFTGenerator(hc, dtype=torch.float32, f32_products="exact").load_state_dict(hift_state_dict(hc))
    t_hift_x = ev_time(lambda: hift_x.inference(speech_feat=mel, cache_source=zero), 4, warm=2)
    res["stages_ms"] = {"note": "one batch of 8 utterances, stages run back to back on all 256 CUs (in the pipeline they overlap on disjoint CU sets)",
                        "llm_prefill": round(t_pre * 1e3, 2), "llm_decode_249_steps": round((t_llm - t_pre) * 1e3, 2),
                        "flow_encoder": round(t_enc * 1e3, 2), "flow_solver_10_steps": round((t_flow - t_enc) * 1e3, 2),
                        "hift_bf16x3": round(t_hift * 1e3, 2), "hift_exact_f32": round(t_hift_x * 1e3, 2)}
    del hift_x
    # ---------------- rooflines
    tf_flow = 4.937 * B   # estimator flops per utterance at T = 1000, 10 steps
    res["roofline_flow"] = {"bound": "mfma", "what": "flow stage (encoder + 10 CFG Euler steps), batch 8, T = 1000, all CUs",
                            "achieved": round(tf_flow / t_flow, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tf_flow / t_flow / 2500.0, 4),
                            "flops": tf_flow * 1e12, "ms": round(t_flow * 1e3, 2), "traffic": None}
    res["roofline"] = measure_attention_roofline(flow, B)
    rows = min(llm.max_batch, UTT_PER_GPU * max(1, args.llm_merge))
    dec = measure_decode_roofline(llm, lc, rows=rows)
    step_s = (t_llm - t_pre) / (N_GEN - 1)
    # 2 (P_layers + P_head) weight bytes + KV: B ctx 12 288 + B 12 288, ctx ~ 282 + 125 on average
    step_bytes = 2 * (357_897_216 + 896 + 5_887_908) + B * (407 + 1) * 12288
    dec["step_level"] = {"what": f"whole decode step (124 launches, one hipGraph replay), {B} sequences, all CUs", "bytes_per_step": step_bytes,
                         "us_per_step": round(step_s * 1e6, 1), "achieved": round(step_bytes / step_s / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(step_bytes / step_s / 8e12, 4)}
    res["roofline_decode"] = dec
    # ---------------- C1 on the GPU: zero-shot, 3 s prompt (75 prompt tokens / 150 prompt frames) + 20-token text, one utterance end to end
    # (prefill 107 -> 250 tokens, flow T = 650, HiFT 500 frames, waveform on the host); RTF as the reference logs it (cli/cosyvoice.py:73-74)
    n3 = 75
    c1a = (one["texts"][:1], one["prompt_texts"][:1], [one["llm_prompt_speech_tokens"][0][:, :n3]], one["flow_prompt_speech_tokens"][:1, :n3],
           one["prompt_speech_feats"][:1, :2 * n3], one["flow_embeddings"][:1])
    run_c1 = lambda: model.tts_batch(*c1a, forced=forced_m[:1], to_host=True)
    for _ in range(2):
        run_c1()
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        w1 = run_c1()
        ts.append(time.perf_counter() - t0)
    lat_c1 = statistics.median(ts)
    res["c1_zero_shot_3s_prompt"] = {"workload": "BASELINE C1 on 1 x MI355X: zero-shot, 3 s prompt + 20-token text, 1 utterance, prefill 107, 250 tokens, "
                                                 "flow T = 650, HiFT 500 frames -> 10 s of audio on the host",
                                     "latency_ms": round(lat_c1 * 1e3, 2), "rtf": round(lat_c1 / AUDIO_S_PER_UTT, 5),
                                     "audio_s_per_s": round(AUDIO_S_PER_UTT / lat_c1, 2), "samples": int(w1.shape[-1])}
    # ---------------- C2: flow-matching decoder only, one utterance, 80 mel x 500 frames, 10 Euler steps
    tok1 = tok[:1]
    ptok0 = torch.zeros(1, 0, dtype=torch.int32, device=dev)
    pfeat0 = torch.zeros(1, 0, 80, device=dev)
    c2args = (tok1, ptok0, pfeat0, one["flow_embeddings"][:1])
    t_c2 = ev_time(lambda: flow.inference_batch(*c2args), 5, warm=2)
    res["c2_flow_only"] = {"workload": "C2: flow-matching decoder only, 1 utterance, 250 tokens -> 80 mel x 500 frames, encoder + 10 CFG Euler steps",
                           "dtype": str(fdt).replace("torch.", ""), "ms": round(t_c2 * 1e3, 2), "tflops": round(1.895 / t_c2, 1),
                           "frac_of_2.5PF": round(1.895 / t_c2 / 2500.0, 4),
                           "note": "batch 1: ~4 700 dependent launches in one hipGraph (16-row tiles, 64-query attention workgroups: 126 - 128 workgroups "
                                   "per launch), each ~10 us of dependent chain; bf16 operands miss the 1e-3 mel target (DESIGN.md §2), so the bench default is fp16"}
    # ---------------- C3: HiFT only, v1 22.05 kHz generator, 10 s of mel (861 frames -> 220 416 samples)
    hc1 = HiftConfig.v1()
    hsd1 = hift_state_dict(hc1)
    mel1 = torch.clamp(torch.randn(1, 80, 861, device=dev) * 2 - 6, -11.5, 2.0)
    c3 = {"workload": "C3: HiFT only, v1 22.05 kHz generator, 10 s of mel (861 frames -> 220 416 samples), fp32 tensors",
          "flops": 518.37e9, "compulsory_bytes_fp32": (457.6e6 + 20.46e6) * 4}
    for mode in ("bf16x3", "exact"):
        h1 = HiFTGenerator(hc1, dtype=torch.float32, f32_products=mode).load_state_dict(hsd1)
        t = ev_time(lambda: h1.inference(speech_feat=mel1, cache_source=zero), 6, warm=2)
        mf = 518.37e9 * (3 if mode == "bf16x3" else 1) / t / 1e12   # MFMA work actually issued: three bf16 products per fp32 product
        c3[mode] = {"ms": round(t * 1e3, 3), "rtf": round(t / 10.0, 6),
                    "hbm_frac": round(c3["compulsory_bytes_fp32"] / t / 8e12, 4),
                    "mfma_tflops_issued": round(mf, 1), "mfma_frac": round(mf / (2500.0 if mode == "bf16x3" else 157.3), 4)}
        del h1
    res["c3_hift_only"] = c3
    # ---------------- HiFT roofline (BASELINE C3 asks for the HBM roofline): the pipeline's own vocoder (v2, 24 kHz), one utterance of 500 frames
    mel_1 = mel[:1].contiguous()
    src_1 = torch.zeros(1, 1, mel_1.shape[2] * hc.total_upsample, device=dev)
    t_h1 = ev_time(lambda: hift.decode(mel_1, src_1), 8, warm=3)
    comp = (385.8e6 + 20.82e6) * 4      # sum over convs of (in + out) elements + parameters, fp32
    total, tsrc = None, {"status": "no usable profile"}
    try:
        import hashlib
        d = json.load(open(os.path.join(ROOT, "profiles", HIFT_PMC)))
        ok = all(hashlib.sha256(open(os.path.join(ROOT, "cosyvoice_amd", "csrc", f), "rb").read()).hexdigest() == h for f, h in d["source_sha256"].items())
        tsrc = {"file": "profiles/" + HIFT_PMC, "measured_live": False, "collected_at": d.get("git_head_at_collection"), "units": d.get("units"),
                "status": "current" if ok else "stale: the profile was collected on a different version of the kernel sources"}
        if ok and d.get("units"):
            total = int(d["total_hbm_bytes"] / d["units"])
    except Exception as e:
        tsrc = {"status": f"no usable profile ({type(e).__name__})"}
    res["roofline_hift"] = {"bound": "hbm", "what": "HiFTGenerator.decode, v2 24 kHz, 1 utterance x 500 frames -> 240 000 samples, fp32 tensors, bf16x3 products (~100 conv launches)",
                            "achieved": round(comp / t_h1 / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(comp / t_h1 / 8e12, 4),
                            "compulsory_bytes": comp, "traffic": total, "traffic_source": tsrc, "ms": round(t_h1 * 1e3, 3),
                            "mfma_tflops_issued": round(3 * 306.15e9 / t_h1 / 1e12, 1), "mfma_frac": round(3 * 306.15e9 / t_h1 / 2.5e15, 4),
                            "note": "the decode is MFMA-issue / latency bound, not HBM bound: at 8 TB/s the compulsory 1.63 GB take 0.2 ms"}
    return res


def pmc_traffic(fname, kernel_substr, src):
    """HBM bytes per launch of a kernel from a committed pair of rocprofv3 --pmc passes (tools/pmc_hbm.py) — REPLAYED from the
    profile, not measured by this run, and only when the profile was collected on the kernel source this run is using (the JSON
    records the file's SHA-256).  Returns (bytes or None, provenance dict)."""
    import hashlib
    info = {"file": "profiles/" + fname, "kernel_source": "cosyvoice_amd/csrc/" + src, "measured_live": False}
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", fname)))
        want = hashlib.sha256(open(os.path.join(ROOT, "cosyvoice_amd", "csrc", src), "rb").read()).hexdigest()
    except Exception as e:
        return None, dict(info, status=f"no usable profile ({type(e).__name__})")
    if d.get("source_sha256", {}).get(src) != want:
        return None, dict(info, status="stale: the profile was collected on a different version of the kernel source", collected_at=d.get("git_head_at_collection"))
    for k, v in d.get("kernels", {}).items():
        if kernel_substr in k:
            return v["hbm_bytes_per_launch"], dict(info, status="current", kernel=k, launches=v["launches"], collected_at=d.get("git_head_at_collection"))
    return None, dict(info, status="kernel not in the profile")


def measure_attention_roofline(flow, B):
    """The by-time dominant kernel of the pipeline: the estimator's flash attention (`attn_kernel`, 560 launches per batch =
    ~44 % of the flow stage).  One launch = all 8 heads x 2 B CFG rows at T = 1000: 4 R H T^2 64 flop; 20 launches captured
    once and replayed between HIP events on the stream they run on."""
    from cosyvoice_amd import ops
    est = flow.decoder.estimator
    cfg = flow.cfg
    R, T = 2 * B, 2 * (N_PROMPT + N_GEN)
    ws = est._workspace(R, T)
    H, inner, Tp = cfg.est_heads, cfg.est_inner, ws["Tp"]

    def launch():
        ops.attention(ws["qk"], ws["qk"][:, :, inner:], ws["vt"], ws["ao"], B=R, H=H, Hkv=H, Tq=T, Tk=T, scale=cfg.est_head_dim ** -0.5,
                      q_bs=T * 2 * inner, ldq=2 * inner, k_bs=T * 2 * inner, ldk=2 * inner, vt_ld=Tp, o_bs=T * inner, ldo=inner)
    n = 20
    g = ops.Graph().capture(lambda: [launch() for _ in range(n)])
    dur = ev_time(g.launch, 5, warm=2) / n
    g.destroy()
    flops = 4.0 * R * H * T * T * 64
    alg_bytes = 4 * R * T * H * 64 * 2   # Q, K, V^T read + O written once, 16-bit
    # separate --pmc FETCH_SIZE / WRITE_SIZE passes over tools/tblock_bench.py at this shape (R = 16, T = 1000)
    traffic, tsrc = pmc_traffic(ATTN_PMC, "attn_kernel<2, 4>", "attention.hip") if (R, T) == (16, 1000) else (None, {"status": "no profile for this shape"})
    return {"bound": "mfma", "kernel": f"attn_kernel<f16, 4 waves> (flow estimator flash attention, {R} rows x {H} heads x T={T}, head_dim 64)",
            "achieved": round(flops / dur / 1e12, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(flops / dur / 2.5e15, 4),
            "traffic": traffic, "traffic_source": tsrc, "algorithmic_bytes_per_launch": alg_bytes, "flops_per_launch": flops, "avg_launch_us": round(dur * 1e6, 2),
            "note": "algorithmic flops per launch = 4 R H T^2 d (QK^T and PV), SURVEY.md §8d's 2048 T flop per row per block; traffic = HBM bytes per "
                    "launch from PMC (2 x FETCH_SIZE + WRITE_SIZE): every (row, head)'s K / V^T is fetched once (one XCD per head)"}


def measure_decode_roofline(llm, lc, rows=UTT_PER_GPU):
    """The decode step's largest weight stream: the gate/up skinny GEMM (exactly one shape in this workload, so the rocprof
    per-kernel average and this live measurement describe the same launches): it streams the layer's 2*4864*896 bf16
    gate/up weights once per launch for all ``rows`` sequences of a token loop.  Algorithmic bytes per launch: DESIGN.md §6."""
    from cosyvoice_amd import ops
    st, lay = llm.st, llm.layers
    H, I = lc.hidden_size, lc.intermediate_size
    B = rows
    split = llm.split_norm
    npart = st["ssp"].shape[0]

    def launch(l):
        if split:   # the form the decode step runs: 16-bit rows + the producer's partial sums of squares, 1/rms in the epilogue
            ops.skinny_gemm(st["xb"], l["p_gu_g"], B, 2 * I, H, mode=2, out_act=st["h"], ldoa=I,
                            split_in=dict(rs=st["ssp"], n=npart, eps=lc.rms_eps))
        else:
            ops.skinny_gemm(st["xn"], l["p_gu"], B, 2 * I, H, mode=2, out_act=st["h"], ldoa=I,
                            norm=dict(x=st["x2"], gamma=l["g_post"], eps=lc.rms_eps))

    # the 24 layers' launches (417 MB of weights > Infinity Cache) are captured once and replayed between HIP events
    g = ops.Graph().capture(lambda: [launch(l) for l in lay])
    dur = ev_time(g.launch, 50, warm=2) / len(lay)
    g.destroy()
    if split:
        alg = 2 * I * H * 2 + B * H * 2 + npart * (16 if B <= 16 else 32) * 4 + B * I * 2   # packed bf16 weights + bf16 rows + partial sums + bf16 SwiGLU out
    else:
        alg = 2 * I * H * 2 + B * H * 4 + H * 4 + B * I * 2   # packed bf16 weights + fp32 residual rows + gamma + bf16 SwiGLU out
    # HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/roofline_pmc.py <rows> split
    traffic, tsrc = pmc_traffic(DECODE_PMC % B, "skinny_kernel<", "llm.hip") if split else (None, {"status": "no profile for the fused-prologue form"})
    return {"bound": "hbm", "kernel": (f"skinny_kernel<{'f16' if llm.dtype == torch.float16 else 'bf16'},TPW=2,no prologue,U=7,RS{',MR=2' if B > 16 else ''}> (decode gate/up, split RMSNorm: 1/rms in the epilogue, + SwiGLU), {B} rows" if split else
                       f"skinny_kernel<{'f16' if llm.dtype == torch.float16 else 'bf16'},TPW=2,norm,TPR={32 if B <= 8 else 16},U=7> (decode gate/up + RMSNorm prologue + SwiGLU), {B} rows"),
            "achieved": round(alg / dur / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(alg / dur / 8e12, 4),
            "traffic": traffic, "traffic_source": tsrc, "bytes_per_launch": alg, "avg_launch_us": round(dur * 1e6, 3)}


if __name__ == "__main__":
    main()
