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


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_pack_kernel_on_synthetic_receive_buffers(gpu_device, world):
    """gf_pack_gathered_hits_device (the device half of gf_allgather_hits_device) on receive buffers as the
    all-gather delivers them for 1, 2, 3 and 8 ranks: ragged counts, an empty rank, a rank over capacity."""
    from genefuserust_amd.dist import pack_gathered_hits
    dev = torch.device("cuda", gpu_device)
    rng = np.random.default_rng(world)
    cap = 700
    counts = [int(c) for c in rng.integers(0, cap + 1, size=world)]
    if world >= 2:
        counts[1] = 0
    if world >= 3:
        counts[2] = cap
    recv = torch.full((world * (cap + 1), HIT_WORDS), -7, dtype=torch.int64)
    want, base = [], 0
    for r, c in enumerate(counts):
        blk = recv[r * (cap + 1):(r + 1) * (cap + 1)]
        blk[0] = 0
        blk[0, 0] = c
        rows = torch.arange(base, base + c, dtype=torch.int64)[:, None] * 10 + torch.arange(HIT_WORDS)[None, :]
        blk[1:1 + c] = rows
        want.append(rows)
        base += c
    merged, totals = pack_gathered_hits(recv.to(dev), world, cap)
    tot = totals.cpu().tolist()
    assert tot[0] == sum(counts) and tot[1] == 0 and tot[2:] == counts
    assert torch.equal(merged[: tot[0]].cpu(), torch.cat(want))
    # a rank over capacity: its block is cut at cap and the flag says so
    recv[0, 0] = cap + 5
    merged, totals = pack_gathered_hits(recv.to(dev), world, cap)
    tot = totals.cpu().tolist()
    assert tot[1] == 1 and tot[2] == cap + 5 and tot[0] == cap + sum(counts[1:])


@pytest.mark.gpu
def test_rccl_exchange_through_the_c_abi_one_rank(gpu_device):
    """gf_comm_unique_id / gf_comm_init / gf_allgather_hits_device / gf_comm_free with RCCL itself, one rank (a test
    box has one GPU): stage, ncclAllGather and pack on a side stream, two batches in flight, the overflow flag."""
    from genefuserust_amd.dist import RcclHitExchange
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dev = torch.device("cuda", gpu_device)
    dist.init_process_group("gloo", rank=0, world_size=1)   # carries the id only
    try:
        hits, n_hits = _fake_hits(0, 5000, cap=6000)
        ex = RcclHitExchange(cap=2048, device=dev)
        h1 = ex.start(hits.to(dev), n_hits.to(dev))
        h2 = ex.start(hits[:10].to(dev), torch.tensor([7], device=dev))
        merged, totals = ex.finish(h1)
        t = totals.cpu().tolist()
        assert t[0] == int(n_hits) and t[1] == 0 and t[2] == int(n_hits)
        assert torch.equal(merged[: t[0]].cpu(), hits[: int(n_hits)])
        merged, totals = ex.finish(h2)
        assert int(totals[0]) == 7 and torch.equal(merged[:7].cpu(), hits[:7])
        assert ex.comm_world() == (0, 1)   # what RCCL reports from inside the communicator (bench.py: ranks_seen)
        small = RcclHitExchange(cap=100, device=dev)
        # the first exchange as bench.py runs it: queued, agreed upon over the torch group, then waited for
        _, totals = small.first_exchange(hits.to(dev), n_hits.to(dev))
        assert int(totals[1]) == 1 and int(totals[0]) == 100
        small.close()
        ex.close()
    finally:
        dist.destroy_process_group()


def _csv_hits(csv, lo, hi, cap):
    """Ordered hit records of reads lo..hi-1 against CSV `csv`: another rule per CSV."""
    ids = [r for r in range(lo, hi) if (r + 3 * csv) % (5 + 2 * csv) == 1]
    h = torch.zeros((cap, HIT_WORDS), dtype=torch.int64)
    for k, r in enumerate(ids):
        h[k, 0] = r
        h[k, 1] = 1 + ((r + csv) % 2)
        h[k, 3] = 1000 * csv + r
    return h, torch.tensor([len(ids)], dtype=torch.int64)


def _group_worker(rank, world, port, n_csv, n, q):
    from genefuserust_amd.multi_csv import plan_multi_csv
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        inner = world // n_csv
        groups = {}
        for k in range(n_csv):   # created collectively, in the same order on every rank (bench.py::bench_multi_csv)
            g = tuple(range(k * inner, (k + 1) * inner))
            groups[g] = dist.new_group(list(g))
        jobs = plan_multi_csv(n_csv, n, rank, world)
        assert len(jobs) == (1 if rank < n_csv * inner else 0)
        for j in jobs:
            assert rank in j.group and len(j.group) == inner and j.group in groups
            hits, n_hits = _csv_hits(j.csv, j.lo, j.hi, cap=j.hi - j.lo + 1)
            merged = allgather_hits(hits, n_hits, group=groups[j.group])
            ex = HitExchange(cap=(n // inner) // 2 + 8, device="cpu", group=groups[j.group])
            ha = ex.start(hits, n_hits)
            hb = ex.start(hits[:3], torch.tensor([min(3, int(n_hits))], dtype=torch.int64))   # a second batch in flight
            m1, t1, o1 = ex.finish(ha)
            m2, t2, o2 = ex.finish(hb)
            assert not bool(o1) and not bool(o2)
            q.put((rank, j.csv, j.lo, j.hi, merged.numpy().copy(), m1[: int(t1)].numpy().copy(), int(t2)))
    finally:
        dist.destroy_process_group()


def test_multi_csv_groups_of_four_ranks_gloo():
    """BASELINE configs[4] with fewer CSVs than ranks — 8 ranks, 2 CSVs: plan_multi_csv forms two groups of four
    (fusion_scan.rs:103-110 with ranks for threads), each rank maps a quarter of the reads against its group's CSV, the
    group merges its lists with the path's one exchange INSIDE the group (both forms: allgather_hits and HitExchange
    with two batches in flight).  Every rank of a group ends with the list one process produces for that CSV."""
    world, n_csv, n = 8, 2, 4001
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_group_worker, args=(r, world, port, n_csv, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(g[0] for g in got) == list(range(world))
    for rank, csv, lo, hi, merged, ex_merged, t2 in got:
        assert csv == rank // 4
        pos = rank % 4
        assert (lo, hi) == shard_range(n, pos, 4)
        want, cnt = _csv_hits(csv, 0, n, cap=n)
        want = want[: int(cnt)].numpy()
        assert np.array_equal(merged, want) and np.array_equal(ex_merged, want)
        assert (np.diff(merged[:, 0]) > 0).all()
        # the second batch: every rank of the group sent its first min(3, count) records
        assert t2 == sum(min(3, int(_csv_hits(csv, *shard_range(n, p, 4), cap=n)[1])) for p in range(4))


def _rccl_worker(rank, world, port, q):
    """One rank of the two-GPU RCCL test: its own device, torch's nccl group for the id and the agreements, the exchange
    through the C ABI."""
    from genefuserust_amd.dist import RcclHitExchange
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        n = 9001
        lo, hi = shard_range(n, rank, world)
        hits, n_hits = _fake_hits(lo, hi, cap=hi - lo + 1)   # ragged: the shards' counts differ
        ex = RcclHitExchange(cap=2048, device=dev)
        assert ex.comm_world() == (rank, world)
        merged, totals = ex.first_exchange(hits.to(dev), n_hits.to(dev))
        t = totals.cpu().tolist()
        q.put((rank, "first", merged[: t[0]].cpu().numpy().copy(), t))
        # two batches in flight: the whole list, then only this rank's first five records
        ha = ex.start(hits.to(dev), n_hits.to(dev))
        hb = ex.start(hits[:5].to(dev), torch.tensor([5], device=dev))
        for tag, h in (("a", ha), ("b", hb)):
            merged, totals = ex.finish(h)
            torch.cuda.synchronize(dev)
            t = totals.cpu().tolist()
            q.put((rank, tag, merged[: t[0]].cpu().numpy().copy(), t))
        small = RcclHitExchange(cap=16, device=dev)           # over capacity on every rank: reported, not cut silently
        _, totals = small.first_exchange(hits.to(dev), n_hits.to(dev))
        q.put((rank, "small", None, totals.cpu().tolist()))
        small.close()
        ex.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_rccl_exchange_two_ranks_on_two_gpus():
    """gf_allgather_hits_device between two ranks over RCCL — ragged counts, two batches in flight, overflow — on a box
    with at least two GPUs.  The pool's boxes have one: there this test is skipped, and the first run of
    the exchange over xGMI is the driver's scaling run."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    world, n = 2, 9001
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rccl_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(4 * world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want, cnt = _fake_hits(0, n, cap=n)
    want = want[: int(cnt)].numpy()
    per_rank = [int(_fake_hits(*shard_range(n, r, world), cap=n)[1]) for r in range(world)]
    for rank, tag, merged, t in got:
        if tag in ("first", "a"):
            assert t[0] == int(cnt) and t[1] == 0 and t[2:2 + world] == per_rank
            assert np.array_equal(merged, want)
        elif tag == "b":
            assert t[0] == 5 * world and t[1] == 0
            firsts = np.concatenate([_fake_hits(*shard_range(n, r, world), cap=n)[0][:5].numpy() for r in range(world)])
            assert np.array_equal(merged, firsts)
        else:
            assert t[1] == 1 and t[0] == 16 * world and t[2:2 + world] == per_rank
