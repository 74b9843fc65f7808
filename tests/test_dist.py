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

from genefuserust_amd.dist import HIT_WORDS, allgather_hits, shard_range


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
    got = [q.get(timeout=120) for _ in range(3 * world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want, cnt = _fake_hits(0, n, cap=n)
    want = want[: int(cnt)].numpy()
    by_rank = {r: [g for rr, g in got if rr == r] for r in range(world)}
    lo1, hi1 = shard_range(n, 1, world)
    w2, c2 = _fake_hits(lo1, hi1, cap=hi1 - lo1 + 1)
    for r in range(world):
        first, second, third = by_rank[r]
        assert np.array_equal(first, want)            # same list as one process, ascending read id
        assert (np.diff(first[:, 0]) > 0).all()
        assert np.array_equal(second, w2[: int(c2)].numpy())
        assert third.shape == (0, HIT_WORDS)
