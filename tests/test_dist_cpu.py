"""CPU, world_size 2, gloo: the N>1 plumbing of the utterance-parallel path (sharding, the single conditioning
broadcast, MAX-over-ranks timing)."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cosyvoice_amd import dist as cd
    g = torch.Generator().manual_seed(123)  # the "true" conditioning, known to every rank for checking
    pfeat = torch.randn(1, 40, 80, generator=g)
    emb = torch.randn(1, 192, generator=g)
    ps = torch.randint(0, 6561, (1, 20), generator=g, dtype=torch.int32)
    pt = torch.randint(0, 151936, (1, 7), generator=g, dtype=torch.int32)
    buf, layout = cd.pack_conditioning(pfeat, emb, ps, pt)
    if rank != 0:
        buf = torch.zeros_like(buf)  # only rank 0 holds the payload before the broadcast
    cd.broadcast_conditioning(buf, dist, src=0)
    a, b, c, d = cd.unpack_conditioning(buf, layout)
    ok = torch.equal(a, pfeat) and torch.equal(b, emb) and torch.equal(c, ps) and torch.equal(d, pt)
    mine = cd.shard_utterances(7, world, rank)
    t = cd.max_over_ranks(1.0 + rank, torch.device("cpu"), dist)
    q.put((rank, ok, mine, t))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_broadcast_shard_and_max():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res)
    assert res[0][2] == [0, 2, 4, 6] and res[1][2] == [1, 3, 5]
    assert all(abs(r[3] - 2.0) < 1e-12 for r in res)  # MAX over ranks


def _ring_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cosyvoice_amd import dist as cd

    def payload(i):   # batch i's conditioning: known to every rank only for checking, held by rank 0 only for sending
        g = torch.Generator().manual_seed(500 + i)
        return (torch.randn(1, 12, 80, generator=g), torch.randn(1, 192, generator=g),
                torch.randint(0, 6561, (1, 6), generator=g, dtype=torch.int32), torch.randint(0, 151936, (1, 5), generator=g, dtype=torch.int32))
    _, layout = cd.pack_conditioning(*payload(0))
    ring = cd.ConditioningRing(3, layout, torch.device("cpu"))
    held, ok, exhausted = [], True, False
    for i in range(7):
        if len(held) == 3:                      # pipeline depth reached: the OLDEST batch finishes and frees its slot
            j, slot, views = held.pop(0)
            pf, em, ps, pt = payload(j)         # ... and until then nothing has touched its tensors
            ok &= all(torch.equal(a, b) for a, b in zip(views, (pf, em, ps, pt)))
            ring.release(slot)
        slot = ring.acquire()
        if rank == 0:
            ring.slots[slot].copy_(cd.pack_conditioning(*payload(i))[0])
        else:
            ring.slots[slot].fill_(-1.0)        # stale contents of a recycled slot must be replaced by the broadcast
        cd.broadcast_conditioning(ring.slots[slot], dist, src=0)
        ring.after_broadcast(slot)
        held.append((i, slot, ring.tensors(slot)))
    try:
        ring.acquire()
    except RuntimeError:
        exhausted = True
    q.put((rank, ok, exhausted, ring.high_water, ring.in_use()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_conditioning_ring():
    """Per-batch conditioning through cosyvoice_amd.dist.ConditioningRing with 2 gloo ranks: distinct payload per batch, one
    broadcast each, a slot is recycled only after its batch is done, and a batch's tensors stay intact while later broadcasts run."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_ring_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, exhausted, high, in_use in res:
        assert ok and exhausted and high == 3 and in_use == 3


def test_token_ids_exact_in_fp32_payload():
    from cosyvoice_amd import dist as cd
    ids = torch.tensor([[0, 151935, 6560, 2 ** 24 - 1]], dtype=torch.int32)
    buf, layout = cd.pack_conditioning(torch.zeros(1, 2, 80), torch.zeros(1, 192), ids, ids)
    _, _, c, d = cd.unpack_conditioning(buf, layout)
    assert torch.equal(c, ids) and torch.equal(d, ids)
