"""Multi-GPU plumbing for the utterance-parallel path: one process per GPU, utterances sharded
round-robin, ONE broadcast per batch of the shared prompt conditioning from rank 0 (RCCL over xGMI on the GPU box,
gloo in the CPU tests), no data-path collective otherwise; timing = MAX over ranks.

The payload is one flat fp32 buffer [prompt mel (T_p*80) | speaker embedding (D) | prompt speech tokens (N_p) |
prompt text ids (L_pt)] (token ids < 2^24 are exact in fp32) — <= ~165 KB, latency-bound."""
from typing import List, Tuple

import torch


def shard_utterances(n_utts: int, world: int, rank: int) -> List[int]:
    """rank r takes utterances {i : i mod world = r}."""
    return list(range(rank, n_utts, world))


def pack_conditioning(prompt_feat: torch.Tensor, embedding: torch.Tensor, prompt_speech: torch.Tensor, prompt_text: torch.Tensor):
    parts = [prompt_feat.reshape(-1).float(), embedding.reshape(-1).float(), prompt_speech.reshape(-1).float(),
             prompt_text.reshape(-1).float()]
    layout = (tuple(prompt_feat.shape), tuple(embedding.shape), tuple(prompt_speech.shape), tuple(prompt_text.shape))
    return torch.cat(parts), layout


def unpack_conditioning(buf: torch.Tensor, layout) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    out, off = [], 0
    for i, shape in enumerate(layout):
        n = 1
        for s in shape:
            n *= s
        t = buf[off:off + n].view(shape)
        out.append(t if i < 2 else t.to(torch.int32))
        off += n
    return tuple(out)


def broadcast_conditioning(buf: torch.Tensor, dist=None, src: int = 0):
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(buf, src=src)
    return buf


def max_over_ranks(value: float, device, dist=None) -> float:
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
