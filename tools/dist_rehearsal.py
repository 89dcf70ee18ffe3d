"""Rehearsal of the utterance-parallel path on ONE GPU: W ranks (torchrun, --backend gloo: RCCL refuses two ranks
on one device) run the bench's multi-process schedule — global utterance list sharded with cosyvoice_amd.dist.shard_utterances, one
broadcast per batch of that batch's OWN conditioning payload into a ConditioningRing slot, CosyVoice2Model.tts_batches — and record
a SHA-256 of every utterance's mel (the flow consumes the broadcast prompt mel / embedding / prompt tokens; the vocoder's source noise
comes from torch's global generator, so waveforms are compared by length only).  Run with WORLD_SIZE=1 and --emulate-world W the ONE
rank walks the W shards one after the other with the same batch composition: the two runs' digests must be identical.

    python tools/dist_rehearsal.py --emulate-world 2 --out gpurun_out/rehearsal_w1.json
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \\
        tools/dist_rehearsal.py --backend gloo --out gpurun_out/rehearsal_w2.json
    python tools/dist_rehearsal.py --compare gpurun_out/rehearsal_w1.json gpurun_out/rehearsal_w2.json
"""
import argparse
import hashlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--emulate-world", type=int, default=0)
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--out", default=None)
    ap.add_argument("--compare", nargs=2, default=None)
    args = ap.parse_args()
    if args.compare:
        a, b = (json.load(open(f)) for f in args.compare)
        assert a["mel_sha256"] == b["mel_sha256"], "mel digests differ between the two runs"
        assert a["wav_samples"] == b["wav_samples"]
        print(f"rehearsal: {len(a['mel_sha256'])} utterance mels identical between {a['world']}-rank and {b['world']}-rank runs "
              f"({a['mode']} vs {b['mode']}); ring high water {a['ring_high_water']} / {b['ring_high_water']}")
        return

    import bench as B
    from cosyvoice_amd import dist as cd
    from cosyvoice_amd import ops
    from cosyvoice_amd.config import FlowConfig, HiftConfig, LlmConfig
    from cosyvoice_amd.flow import CausalMaskedDiffWithXvec
    from cosyvoice_amd.hift import HiFTGenerator
    from cosyvoice_amd.llm import Qwen2LM
    from cosyvoice_amd.model import CosyVoice2Model
    from cosyvoice_amd.weights import flow_state_dict, hift_state_dict, llm_state_dict

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    torch.set_num_threads(8)
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend)
    W = world if world > 1 else max(1, args.emulate_world)
    shards = [rank] if world > 1 else list(range(W))

    lc, fc, hc = LlmConfig.full(), FlowConfig.full(), HiftConfig.v2()
    llm = Qwen2LM(lc, dtype=torch.float16, max_batch=16, ctx_max=704, max_out=B.N_GEN + 8)
    flow = CausalMaskedDiffWithXvec(fc, dtype=torch.float16)
    hift = HiFTGenerator(hc, dtype=torch.float32)
    model = CosyVoice2Model(llm, flow, hift, fp16=False).load_state_dicts(llm_state_dict(lc), flow_state_dict(fc), hift_state_dict(hc))
    model.llm_merge = 2
    model.flow_merge = int(os.environ.get("REHEARSAL_FLOW_MERGE", "2"))   # flow groups of two batches: the merged pass the bench runs
    flow.decoder.use_graph = True
    dev = torch.device("cuda")
    n_global = B.UTT_PER_GPU * W
    texts, forced, ptext, pspeech, pfeat, emb = B.make_inputs(lc, fc, seed=100, n_utts=n_global)
    _, layout = cd.pack_conditioning(pfeat, emb, pspeech, ptext)

    def payload_of(i):
        g = torch.Generator().manual_seed(7000 + i)
        pf = torch.clamp(pfeat + 0.05 * torch.randn(pfeat.shape, generator=g), -11.5, 2.0)
        em = emb + 0.1 * torch.randn(emb.shape, generator=g)
        ps = torch.randint(0, lc.speech_token_size, pspeech.shape, generator=g, dtype=torch.int32)
        return cd.pack_conditioning(pf, em, ps, ptext)[0]
    payloads = [payload_of(i).to(dev) for i in range(args.steps)] if rank == 0 else None
    ring = cd.ConditioningRing(12, layout, dev)
    digests, samples, order = {}, {}, []
    orig = flow.inference_batch

    def recording(*a, **k):
        mel = orig(*a, **k)
        row = 0
        while row < mel.shape[0]:     # a merged flow pass (flow_merge) covers several consecutive batches, rows in batch order
            step, ids = order.pop(0)
            for u in ids:
                digests[f"{step}:{u}"] = hashlib.sha256(mel[row].float().cpu().numpy().tobytes()).hexdigest()
                row += 1
        return mel
    flow.inference_batch = recording

    t0 = time.time()
    for r in shards:
        mine = cd.shard_utterances(n_global, W, r)

        def batches():
            for i in range(args.steps):
                slot = ring.acquire()
                buf = ring.slots[slot]

                def bcast(i=i, slot=slot, buf=buf):
                    if rank == 0:
                        buf.copy_(payloads[i])
                    cd.broadcast_conditioning(buf, dist, src=0)
                    ring.after_broadcast(slot)
                pf, em, ps, pt = ring.tensors(slot)
                n = len(mine)
                order.append((i, mine))
                yield dict(texts=[texts[u].to(dev) for u in mine], prompt_texts=[pt] * n, llm_prompt_speech_tokens=[ps] * n,
                           flow_prompt_speech_tokens=ps.expand(n, -1), prompt_speech_feats=pf.expand(n, -1, -1),
                           flow_embeddings=em.expand(n, -1), forced=[forced[u] for u in mine], on_start=bcast,
                           on_done=lambda slot=slot: ring.release(slot))
        for i, wav in enumerate(model.tts_batches(batches(), to_host=True)):
            for j, u in enumerate(mine):
                samples[f"{i}:{u}"] = int(wav[j].shape[-1])
            assert torch.isfinite(wav).all()
        assert ring.in_use() == 0
    res = {"mel_sha256": digests, "wav_samples": samples}
    if dist is not None:
        gathered = [None] * world
        dist.all_gather_object(gathered, res)
        res = {"mel_sha256": {k: v for g in gathered for k, v in g["mel_sha256"].items()},
               "wav_samples": {k: v for g in gathered for k, v in g["wav_samples"].items()}}
    if rank == 0:
        res = {"world": world, "mode": (f"{world} ranks, backend {args.backend}" if world > 1 else f"1 rank walking {W} shards"),
               "steps": args.steps, "utterances": n_global * args.steps, "ring_high_water": ring.high_water,
               "seconds": round(time.time() - t0, 2),
               "mel_sha256": dict(sorted(res["mel_sha256"].items())), "wav_samples": dict(sorted(res["wav_samples"].items()))}
        assert len(res["mel_sha256"]) == n_global * args.steps, len(res["mel_sha256"])
        assert len(set(res["mel_sha256"].values())) == len(res["mel_sha256"])   # every (step, utterance) has its own conditioning / tokens
        if args.out:
            json.dump(res, open(args.out, "w"), indent=1)
        print(f"rehearsal [{res['mode']}]: {len(res['mel_sha256'])} utterance mels in {res['seconds']} s, ring high water {ring.high_water}", flush=True)
    if dist is not None:
        dist.barrier()
    model.close()
    ops.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
