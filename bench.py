#!/usr/bin/env python3
"""bench.py — audio-seconds per wall-second of the LLM -> flow -> HiFT hot path on MI355X.

Workload (BASELINE.json configs[3], "Full LLM->flow->hifigan pipeline, batch 8 x 10 s utterances, 1xMI355X with
hipGraph token loop"): per GPU 8 utterances, each a 10 s prompt (N_p=250 prompt speech tokens,
T_p=500 prompt mel frames, 10 prompt-text + 20 text ids -> prefill length 282) generating N_g=250 teacher-forced
speech tokens = 10.0 s of 24 kHz audio; flow T = 1000 frames, 10 CFG Euler steps; HiFT 500 frames -> 240 000 samples.
One "step" = the whole batch through the pipeline (prefill + 249 graph-replayed decode steps incl. on-device sampling,
flow encoder + solver, HiFT, waveform D2H).  The K timed steps are software-pipelined over two HIP streams (LLM decode
of step i+1 overlaps flow + HiFT of step i, as the reference overlaps its LLM thread with flow/HiFT); all K steps' work,
fill and drain included, lies inside the timed region.  Synthetic inputs + key-seeded random weights of the reference's
architecture (no checkpoint exists offline); inputs are resident in HBM before the timed region.

N>1: one process per GPU (torchrun), utterances sharded 8 per rank (weak scaling), ONE RCCL broadcast per step of the
shared prompt conditioning from rank 0; barrier + synchronize bracket the timed region, MAX over ranks.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

UTT_PER_GPU = 8
N_PROMPT, N_GEN, L_TEXT, L_PTEXT = 250, 250, 20, 10
AUDIO_S_PER_UTT = N_GEN * 2 * 480 / 24000.0  # 10.0 s


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def make_inputs(lc, fc, seed):
    g = torch.Generator().manual_seed(seed)
    B = UTT_PER_GPU
    texts = [torch.randint(0, lc.vocab_size, (1, L_TEXT), generator=g, dtype=torch.int32) for _ in range(B)]
    forced = [torch.randint(0, lc.speech_token_size, (N_GEN,), generator=g).tolist() for _ in range(B)]
    # shared prompt conditioning (one speaker prompt for the whole batch: what the broadcast carries)
    ptext = torch.randint(0, lc.vocab_size, (1, L_PTEXT), generator=g, dtype=torch.int32)
    pspeech = torch.randint(0, lc.speech_token_size, (1, N_PROMPT), generator=g, dtype=torch.int32)
    pfeat = torch.clamp(torch.randn(1, 2 * N_PROMPT, 80, generator=g) * 2 - 6, -11.5, 2.0)
    emb = torch.randn(1, fc.spk_embed_dim, generator=g)
    return texts, forced, ptext, pspeech, pfeat, emb


def cpu_baseline(lsd, fsd, hsd, lc, fc, hc):
    """The oracle (CPU port of the reference arithmetic) timed on this host's cores on a bounded sample of the SAME
    workload, extrapolated linearly in the repeated units (decode steps, Euler steps, mel frames)."""
    from oracle import flow as of
    from oracle import hift as oh
    from oracle import llm as ol
    # the GPU box gives one GPU a 16-core CPU share; torch's default (all 128+ hardware threads) is far slower on these
    # small ops than 16 threads, so the baseline is timed at its best setting
    cores = min(16, torch.get_num_threads())
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(1)
    with torch.inference_mode():
        # LLM: prefill L=282 + 6 decode steps
        x = torch.randn(1, 1 + L_PTEXT + L_TEXT + 1 + N_PROMPT, lc.hidden_size, generator=g) * 0.02
        cache = ol.KVCache(lc.num_layers)
        t0 = time.perf_counter(); y = ol.qwen2_forward(lsd, lc, x, cache); ol.logits_to_logp(lsd, y[:, -1]); t_pre = time.perf_counter() - t0
        x1 = torch.randn(1, 1, lc.hidden_size, generator=g) * 0.02
        ol.qwen2_forward(lsd, lc, x1, cache)
        t0 = time.perf_counter()
        for _ in range(6):
            y = ol.qwen2_forward(lsd, lc, x1, cache); ol.logits_to_logp(lsd, y[:, -1])
        t_step = (time.perf_counter() - t0) / 6
        # flow: encoder N=500 once + ONE estimator call at T=1000 (CFG batch 2)
        N = N_PROMPT + N_GEN
        xs = torch.randn(1, N, fc.input_size, generator=g)
        t0 = time.perf_counter(); of.encoder_forward(fsd, fc, xs, torch.tensor([N]), 0); t_enc = time.perf_counter() - t0
        T = 2 * N
        a = lambda *s: torch.randn(*s, generator=g)
        t0 = time.perf_counter()
        of.estimator_forward(fsd, fc, a(2, 80, T), torch.ones(2, 1, T), a(2, 80, T), torch.tensor([0.5, 0.5]), a(2, 80), a(2, 80, T))
        t_est = time.perf_counter() - t0
        # HiFT: 100 of the 500 frames
        mel = torch.clamp(a(1, 80, 100) * 2 - 6, -11.5, 2.0)
        ph, nz = oh.draw_source_randoms(hc, 1, 100 * hc.total_upsample, seed=2)
        t0 = time.perf_counter(); oh.inference(hsd, hc, mel, None, ph, nz); t_hift = (time.perf_counter() - t0) * 5
    per_utt = t_pre + (N_GEN - 1) * t_step + t_enc + fc.n_timesteps * t_est + t_hift
    return {"value": round(AUDIO_S_PER_UTT / per_utt, 4), "unit": "audio-seconds/sec", "cores": cores, "kind": "port",
            "sample": (f"oracle fp32, 1 utterance of the same workload: prefill L=282 ({t_pre:.2f}s) + 6 decode steps "
                       f"({t_step*1e3:.1f} ms/step, x249) + flow encoder N=500 ({t_enc:.2f}s) + 1 estimator call T=1000 "
                       f"({t_est:.2f}s, x10) + HiFT 100/500 frames (x5 = {t_hift:.2f}s); per-utterance {per_utt:.1f}s")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--flow-dtype", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--llm-cu-slots", type=int, default=8,
                    help="CU slots per XCD (of 32) owned by the decode loops while they overlap flow+HiFT; 0 = no partition")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; gloo only to rehearse the multi-process path with several ranks on ONE GPU")
    ap.add_argument("--llm-merge", type=int, default=2,
                    help="consecutive 8-utterance batches decoded by one token loop (2: 16 rows share every weight stream)")
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
    llm = Qwen2LM(lc, dtype=torch.bfloat16, max_batch=UTT_PER_GPU * max(1, args.llm_merge), ctx_max=704, max_out=N_GEN + 8)
    flow = CausalMaskedDiffWithXvec(fc, dtype=fdt)
    hift = HiFTGenerator(hc, dtype=torch.float32)
    model = CosyVoice2Model(llm, flow, hift, fp16=False).load_state_dicts(lsd, fsd, hsd)
    model.llm_merge = max(1, args.llm_merge)
    # the fork drives CosyVoice2 modules through CosyVoiceModel wiring (full attention, model.py:50); the benchmark keeps
    # CosyVoice2Model's own chunk-50 encoder mask (model.py:314)
    flow.decoder.use_graph = True
    log(f"[rank {rank}] models on device in {time.time()-t0:.1f}s")

    texts, forced, ptext, pspeech, pfeat, emb = make_inputs(lc, fc, seed=100 + rank)
    dev = torch.device("cuda", local_rank)
    B = UTT_PER_GPU
    # conditioning buffer broadcast from rank 0: [prompt mel | speaker embedding | prompt speech tokens | prompt text ids]
    n_feat, n_emb = pfeat.numel(), emb.numel()
    cond_buf = torch.cat([pfeat.reshape(-1), emb.reshape(-1), pspeech.reshape(-1).float(), ptext.reshape(-1).float()]).to(dev)
    texts_d = [t.to(dev) for t in texts]

    def bcast():
        if dist is not None:
            dist.broadcast(cond_buf, src=0)

    def make_batch():
        pf = cond_buf[:n_feat].view(1, 2 * N_PROMPT, 80)
        em = cond_buf[n_feat:n_feat + n_emb].view(1, -1)
        ps = cond_buf[n_feat + n_emb:n_feat + n_emb + N_PROMPT].to(torch.int32).view(1, -1)
        pt = cond_buf[n_feat + n_emb + N_PROMPT:].to(torch.int32).view(1, -1)
        return dict(texts=texts_d, prompt_texts=[pt] * B, llm_prompt_speech_tokens=[ps] * B,
                    flow_prompt_speech_tokens=ps.expand(B, -1), prompt_speech_feats=pf.expand(B, -1, -1),
                    flow_embeddings=em.expand(B, -1), forced=forced, on_start=bcast)

    def run_steps(n):
        """n pipeline passes ("steps", one batch of 8 utterances each): the decode loops of later passes (two at a time, on
        their CU share) overlap flow + HiFT of pass i (on the other CUs); every waveform is copied to the host inside the
        region."""
        last = None
        for wav in model.tts_batches([make_batch() for _ in range(n)], to_host=True, llm_cu_slots=args.llm_cu_slots, llm_loops=args.llm_loops):
            last = wav
        return last

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup > 0:
        w = run_steps(args.warmup)
        log(f"[rank {rank}] warmup: wav {tuple(w.shape)} absmax {w.abs().max().item():.3f}")
    fence()
    t0 = time.perf_counter()
    run_steps(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- batch-1 latency of the same pipeline (BASELINE metric: "batch 1 and 8"; north_star target RTF < 0.05), rank 0 only
    b1 = None
    if rank == 0:
        one = dict(texts=texts_d[:1], prompt_texts=[make_batch()["prompt_texts"][0]], llm_prompt_speech_tokens=[make_batch()["llm_prompt_speech_tokens"][0]],
                   flow_prompt_speech_tokens=make_batch()["flow_prompt_speech_tokens"][:1], prompt_speech_feats=make_batch()["prompt_speech_feats"][:1],
                   flow_embeddings=make_batch()["flow_embeddings"][:1], forced=forced[:1])
        def run1():
            return model.tts_batch(one["texts"], one["prompt_texts"], one["llm_prompt_speech_tokens"], one["flow_prompt_speech_tokens"],
                                   one["prompt_speech_feats"], one["flow_embeddings"], forced=one["forced"], to_host=True)
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

    # ---- roofline of the dominant kernel, measured live with events on the launch stream
    roof = measure_roofline(llm, lc, rows=min(llm.max_batch, UTT_PER_GPU * max(1, args.llm_merge)))

    out = None
    if rank == 0:
        audio_s = AUDIO_S_PER_UTT * B * world * args.steps
        out = {
            "metric": "audio-seconds/sec (RTF^-1) end-to-end, 10s prompt, batch 8",
            "value": round(audio_s / elapsed, 3), "unit": "audio-seconds/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": f"llm bf16 / flow {args.flow_dtype} MFMA operands, fp32 accumulate; hift fp32 tensors, bf16x3 split-product MFMA",
            "data": "synthetic (key-seeded random weights of the reference architecture, teacher-forced 250 tokens)",
            "config": {"workload": "C4 full LLM->flow->HiFT pipeline, 8 utterances x 10 s per GPU, 10 s prompt "
                                   "(prefill 282, N_g 250, flow T 1000 x 10 CFG Euler steps, HiFT 500 frames)",
                       "utterances_per_gpu": B, "rtf": round(elapsed / audio_s, 6), "parallelism": f"utterance-parallel x{world}",
                       "llm_cu_slots_per_xcd": args.llm_cu_slots, "llm_decode_loops": args.llm_loops, "batches_per_decode_loop": args.llm_merge, "batch1": b1},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(lsd, fsd, hsd, lc, fc, hc)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def measure_roofline(llm, lc, rows=UTT_PER_GPU):
    """Roofline kernel = the decode step's gate/up skinny GEMM (exactly one shape in this workload, so the rocprof
    per-kernel average and this live measurement describe the same launches): it streams the layer's 2*4864*896 bf16
    gate/up weights once per launch for all ``rows`` sequences of a token loop (8 utterances x the batches one decode loop
    carries) — the largest weight stream of the HBM-bound decode stage (47 % of kernel time).  Algorithmic bytes per launch:
    DESIGN.md §6."""
    from cosyvoice_amd import ops
    st, lay = llm.st, llm.layers
    H, I = lc.hidden_size, lc.intermediate_size
    B = rows
    n_iter = 50
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    # cycle through all 24 layers' weights (417 MB > 256 MB Infinity Cache) so every launch streams from HBM
    split = llm.split_norm
    npart = st["ssp"].shape[0]

    def launch(l):
        if split:   # the form the decode step runs: 16-bit rows + the producer's partial sums of squares, 1/rms in the epilogue
            ops.skinny_gemm(st["xb"], l["p_gu_g"], B, 2 * I, H, mode=2, out_act=st["h"], ldoa=I,
                            split_in=dict(rs=st["ssp"], n=npart, eps=lc.rms_eps))
        else:
            ops.skinny_gemm(st["xn"], l["p_gu"], B, 2 * I, H, mode=2, out_act=st["h"], ldoa=I,
                            norm=dict(x=st["x2"], gamma=l["g_post"], eps=lc.rms_eps))

    # the 24 launches are captured once and replayed: issued one by one from Python the host (~10 us per call) would be
    # the thing timed, not the 7 us kernel.  HIP events bracket the replays on the stream they run on.
    g = ops.Graph().capture(lambda: [launch(l) for l in lay])
    g.launch()
    ev0.record()
    for _ in range(n_iter):
        g.launch()
    ev1.record()
    torch.cuda.synchronize()
    dur = ev0.elapsed_time(ev1) * 1e-3 / (n_iter * len(lay))
    if split:
        alg = 2 * I * H * 2 + B * H * 2 + npart * 16 * 4 + B * I * 2   # packed bf16 weights + bf16 rows + partial sums + bf16 SwiGLU out
    else:
        alg = 2 * I * H * 2 + B * H * 4 + H * 4 + B * I * 2   # packed bf16 weights + fp32 residual rows + gamma + bf16 SwiGLU out
    traffic = None
    pmc = os.path.join(ROOT, "profiles", ("r01_roofline_pmc_split_b%d.json" % B) if split else
                       ("r01_roofline_pmc.json" if B <= 8 else "r01_roofline_pmc_b16.json"))
    if os.path.exists(pmc):  # HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/roofline_pmc.py)
        try:
            traffic = json.load(open(pmc))["hbm_bytes_per_launch"]
        except Exception:
            traffic = None
    note = (f"one launch streams the layer's gate/up weights for {B} sequences ({max(1, B // UTT_PER_GPU)} batch(es) per token loop); the "
            "post-attention RMSNorm is split over the o_proj launch (16-bit rows + per-workgroup partial sums of squares) and this "
            "kernel's epilogue (1/rms; gamma folded into the packed weights), so no prologue sits in front of the MFMAs "
            "(fused-prologue form: 7.4 us at 8 rows, 9.2-9.5 us at 16; tools/roofline_time.py)") if split else None
    return {"note": note, "bound": "hbm", "kernel": (f"skinny_kernel<bf16,TPW=2,no prologue,U=7,RS> (decode gate/up, split RMSNorm: 1/rms in the epilogue, + SwiGLU), {B} rows" if split else
                       f"skinny_kernel<bf16,TPW=2,norm,TPR={32 if B <= 8 else 16},U=7> (decode gate/up + RMSNorm prologue + SwiGLU), {B} rows"),
            "achieved": round(alg / dur / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(alg / dur / 8e12, 4),
            "traffic": traffic, "bytes_per_launch": alg, "avg_launch_us": round(dur * 1e6, 3)}


if __name__ == "__main__":
    main()
