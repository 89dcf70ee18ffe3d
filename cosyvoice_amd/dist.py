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


class ConditioningRing:
    """Device buffers for the per-batch conditioning payload of a pipelined run.  Every batch in flight owns ONE slot from its
    broadcast (``on_start``) until its last consumer has finished (``on_done``, called by CosyVoice2Model.tts_batches once the
    batch's waveform has been collected: its decode, flow and vocoder jobs are complete), so the broadcast of a later batch can never
    rewrite conditioning a running flow / LLM job still reads.  The reference keeps such state per request uuid for the same reason
    (cli/model.py:183-188,374-377).  ``acquire`` raises when every slot is taken: size the ring for the pipeline depth
    (``tts_batches`` admits at most (llm_loops + 2) * llm_merge + llm_merge - 1 batches at once, plus one looked ahead).

    A slot = the flat fp32 payload buffer (what the collective moves) + int32 tensors for the two token fields.  ``tensors(slot)``
    hands out (prompt mel view, embedding view, prompt speech tokens, prompt text ids); the views alias the payload, the token
    tensors are filled from it by ``after_broadcast(slot)`` — stream-ordered behind the broadcast, never at batch-assembly time."""

    def __init__(self, n_slots: int, layout, device):
        import threading
        self.layout = layout
        sizes = []
        for shape in layout:
            n = 1
            for d in shape:
                n *= d
            sizes.append(n)
        self.sizes = sizes
        self.slots = [torch.zeros(sum(sizes), device=device, dtype=torch.float32) for _ in range(n_slots)]
        self.tokens = [(torch.zeros(layout[2], device=device, dtype=torch.int32), torch.zeros(layout[3], device=device, dtype=torch.int32))
                       for _ in range(n_slots)]
        self._free = list(range(n_slots))[::-1]
        self._lock = threading.Lock()
        self.high_water = 0

    def acquire(self) -> int:
        with self._lock:
            if not self._free:
                raise RuntimeError("ConditioningRing exhausted: more batches in flight than slots")
            i = self._free.pop()
            self.high_water = max(self.high_water, len(self.slots) - len(self._free))
            return i

    def release(self, i: int):
        with self._lock:
            assert i not in self._free
            self._free.append(i)

    def in_use(self) -> int:
        with self._lock:
            return len(self.slots) - len(self._free)

    def tensors(self, slot: int):
        buf, (ps, pt) = self.slots[slot], self.tokens[slot]
        n0, n1 = self.sizes[0], self.sizes[1]
        return buf[:n0].view(self.layout[0]), buf[n0:n0 + n1].view(self.layout[1]), ps, pt

    def after_broadcast(self, slot: int):
        """Token ids of the slot's payload -> its int32 tensors (on the current stream, i.e. behind the broadcast)."""
        buf, (ps, pt) = self.slots[slot], self.tokens[slot]
        o = self.sizes[0] + self.sizes[1]
        ps.copy_(buf[o:o + self.sizes[2]].view(self.layout[2]))
        pt.copy_(buf[o + self.sizes[2]:o + self.sizes[2] + self.sizes[3]].view(self.layout[3]))
