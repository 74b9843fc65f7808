"""The N>1 path on CPU: world_size-2 gloo processes shard a batch, 'map' their
shard (hit records synthesised from a deterministic rule — the GPU kernels are
covered by the -m gpu tests), and merge with the product's all-gather.  The
merged list must equal the single-process list, on every rank."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from genefuserust_amd.dist import HIT_WORDS, HitExchange, allgather_hits, shard_range


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_hits(lo, hi, cap):
    """Ordered hit records for reads lo..hi-1: read r 'hits' iff r % 7 == 3 or r % 11 == 0."""
    ids = [r for r in range(lo, hi) if r % 7 == 3 or r % 11 == 0]
    h = torch.zeros((cap, HIT_WORDS), dtype=torch.int64)
    for k, r in enumerate(ids):
        h[k, 0] = r
        h[k, 1] = 1 + (r % 2)
        h[k, 2] = r * 3
        h[k, 5] = -r
    return h, torch.tensor([len(ids)], dtype=torch.int64)


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_range(n, rank, world)
        hits, n_hits = _fake_hits(lo, hi, cap=hi - lo + 1)
        merged = allgather_hits(hits, n_hits)
        q.put((rank, merged.numpy().copy()))
        # second round: one rank has no hits at all, and one round where nobody has
        empty = torch.zeros((4, HIT_WORDS), dtype=torch.int64)
        h2, c2 = (hits, n_hits) if rank == 1 else (empty, torch.zeros(1, dtype=torch.int64))
        q.put((rank, allgather_hits(h2, c2).numpy().copy()))
        q.put((rank, allgather_hits(empty, torch.zeros(1, dtype=torch.int64)).numpy().copy()))
        # the sync-free exchange: two batches in flight, then finished in order
        ex = HitExchange(cap=n // world + 8, device="cpu")   # the same capacity on every rank
        ha = ex.start(hits, n_hits)
        hb = ex.start(h2, c2)
        for h in (ha, hb):
            merged, total, overflow = ex.finish(h)
            assert not bool(overflow)
            q.put((rank, merged[: int(total)].numpy().copy()))
        small = HitExchange(cap=3, device="cpu")   # too small for this rank's list: reported, not silently cut
        _, _, overflow = small.finish(small.start(hits, n_hits))
        assert bool(overflow)
    finally:
        dist.destroy_process_group()


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 1000, 20_000_001):
        for world in (1, 2, 3, 8):
            edges = [shard_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            for a, b in zip(edges, edges[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("world", [2, 3])
def test_allgather_hits_gloo(world):
    n = 1003
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(5 * world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want, cnt = _fake_hits(0, n, cap=n)
    want = want[: int(cnt)].numpy()
    by_rank = {r: [g for rr, g in got if rr == r] for r in range(world)}
    lo1, hi1 = shard_range(n, 1, world)
    w2, c2 = _fake_hits(lo1, hi1, cap=hi1 - lo1 + 1)
    for r in range(world):
        first, second, third, ex_first, ex_second = by_rank[r]
        assert np.array_equal(ex_first, want) and np.array_equal(ex_second, w2[: int(c2)].numpy())
        assert np.array_equal(first, want)            # same list as one process, ascending read id
        assert (np.diff(first[:, 0]) > 0).all()
        assert np.array_equal(second, w2[: int(c2)].numpy())
        assert third.shape == (0, HIT_WORDS)


@pytest.mark.gpu
def test_hit_exchange_on_the_device(gpu_device):
    """The sync-free exchange over RCCL (backend "nccl"), one rank: the CUDA path of the
    packing and of the device-side merge (more ranks need more GPUs than a test box has)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dev = torch.device("cuda", gpu_device)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        hits, n_hits = _fake_hits(0, 5000, cap=6000)
        ex = HitExchange(cap=2048, device=dev)
        h1 = ex.start(hits.to(dev), n_hits.to(dev))
        h2 = ex.start(hits[:10].to(dev), torch.tensor([7], device=dev))
        merged, total, overflow = ex.finish(h1)
        assert not bool(overflow) and int(total) == int(n_hits)
        assert torch.equal(merged[: int(total)].cpu(), hits[: int(n_hits)])
        merged, total, overflow = ex.finish(h2)
        assert int(total) == 7 and torch.equal(merged[:7].cpu(), hits[:7])
        small = HitExchange(cap=100, device=dev)
        _, _, overflow = small.finish(small.start(hits.to(dev), n_hits.to(dev)))
        assert bool(overflow)
    finally:
        dist.destroy_process_group()
