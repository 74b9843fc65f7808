"""BASELINE.json configs[2], [3] and [4] as tests that run on ONE GPU (or none).

configs[2]  100 M pairs vs the cancer-shaped index: the full-size gene set (IDX-C at scale 1.0:
            29 M keys, 0.5 GB table, the filter form for indexes beyond an XCD's L2) with 20 M
            reads, checked like configs[1] through size-independent properties.
configs[3]  reads sharded over 8 ranks + one all-gather: the shards of `shard_range(n, r, 8)`
            mapped one after the other with read_id_base = lo and concatenated in rank order
            must be the one-shot hit list; the same through real gloo processes.
configs[4]  multi-CSV mode: a resident read set, the index rebuilt per CSV (16 alternating
            gene sets), parity per CSV.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from genefuserust_amd.dist import HIT_WORDS, allgather_hits, shard_range
from genefuserust_amd.multi_csv import plan_multi_csv


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _hits_from_dense(counts, matches, base):
    """The ordered gf_hit list (int64[k, 6]) of a dense (counts, matches) result — what K4 writes."""
    from genefuserust_amd._lib import HIT_DTYPE
    idx = np.nonzero(counts)[0]
    h = np.zeros(idx.size, dtype=HIT_DTYPE)
    h["read_id"] = idx + base
    h["n"] = counts[idx]
    for k in (0, 1):
        sel = counts[idx] > k
        for f in ("seq_start", "seq_end", "position", "contig"):
            h["m"][f][sel, k] = matches[f][idx[sel], k]
    return torch.from_numpy(h.view(np.int64).reshape(-1, HIT_WORDS).copy())


# ---------------------------------------------------------------- CPU: plan + exchange of real lists

def test_plan_multi_csv_covers_every_csv_and_read_once():
    for n_csv in (1, 2, 3, 5, 16, 17):
        for world in (1, 2, 3, 4, 8):
            n = 1003
            seen = {}
            for r in range(world):
                for j in plan_multi_csv(n_csv, n, r, world):
                    assert r in j.group
                    seen.setdefault(j.csv, []).append((j.lo, j.hi, j.group))
            assert sorted(seen) == list(range(n_csv)), (n_csv, world)
            for k, parts in seen.items():
                parts.sort()
                assert parts[0][0] == 0 and parts[-1][1] == n
                for a, b in zip(parts, parts[1:]):
                    assert a[1] == b[0] and a[2] == b[2]
                assert len(parts) == len(parts[0][2])
                if n_csv >= world:   # a CSV is one rank's job: no collective
                    assert len(parts) == 1
            # the reference's rule (fusion_scan.rs:103-110): all threads on different CSVs when there are
            # enough CSVs, otherwise threads / n_csv per CSV
            if n_csv < world:
                assert all(len(p[0][2]) == world // n_csv for p in seen.values())


def _oracle_worker(rank, world, port, genes, bases, offsets, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle_py
        ox = oracle_py.OracleIndexer(genes)
        n = offsets.size - 1
        lo, hi = shard_range(n, rank, world)
        c, m = ox.map_reads_packed(bases, offsets[lo:hi + 1], threads=2)   # offsets are absolute: a shard is a slice
        hits = _hits_from_dense(c, m, lo)
        merged = allgather_hits(torch.cat([hits, torch.zeros((3, HIT_WORDS), dtype=torch.int64)]),
                                torch.tensor([hits.shape[0]], dtype=torch.int64))
        q.put((rank, merged.numpy().copy()))
    finally:
        dist.destroy_process_group()


def test_sharded_real_hit_lists_merge_to_the_one_shot_list_gloo(oracle):
    """configs[3] on the CPU: every rank maps its contiguous shard (the oracle stands in for the
    kernels: real SeqMatch records, not synthetic ones), numbers its hits from the shard's first
    read, and the product's all-gather must return the list of the whole batch on every rank."""
    from genefuserust_amd import synth
    genes = synth.make_geneset("IDX-T", scale=0.02)
    synth.MIXES["TEST"] = (0.2, 0.5, 0.3)
    rb = synth.make_reads(genes, 3001, read_len=150, mix="TEST", seed=5)
    bases, offsets = rb.bases.numpy(), rb.offsets.numpy()
    ox = oracle.OracleIndexer(genes.seqs)
    c, m = ox.map_reads_packed(bases, offsets, threads=4)
    want = _hits_from_dense(c, m, 0).numpy()
    assert want.shape[0] > 300
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_oracle_worker, args=(r, world, port, genes.seqs, bases, offsets, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, g in got:
        assert np.array_equal(g, want)


# ---------------------------------------------------------------- GPU

def _full_size_properties(oracle, shape, n, L=150, seed=4242, min_hits=5_000, max_hits=200_000, pairs=False):
    import torch
    from genefuserust_amd import Indexer, synth
    from genefuserust_amd.indexer import hits_to_numpy
    genes = synth.make_geneset(shape)
    ix = Indexer.from_gene_slices(genes.seqs, genes.reversed_flags)
    ix.make_index()
    ox = oracle.OracleIndexer(genes.seqs)
    st, info = ox.stats(), ix.info()
    assert (info["n_keys"], info["n_high_keys"], info["n_unique"]) == (st["n_keys"], st["n_high_keys"], st["m_unique_pos"])
    if pairs:   # the bench's own workload: SURVEY.md 8(d) pairs, R1, R2, R1, R2 ..
        rb = synth.make_pair_reads(genes, n // 2, read_len=L, mix="PANEL", seed=seed, device="cuda")
    else:
        rb = synth.make_reads(genes, n, read_len=L, mix="PANEL", seed=seed, device="cuda")
    counts, matches = ix.map_reads_device(rb.bases, rb.offsets, L)
    torch.cuda.synchronize()
    assert int((counts > 2).sum()) == 0
    hit_idx = torch.nonzero(counts).flatten()
    assert min_hits < hit_idx.numel() < max_hits
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    sample = torch.randint(0, n, (100_000,), device="cuda", generator=gen)
    sel = torch.unique(torch.cat([hit_idx, sample]))
    reads2d = rb.bases.view(n, L)
    sub = reads2d[sel].contiguous().cpu().numpy().reshape(-1)
    offs = np.arange(sel.numel() + 1, dtype=np.int64) * L
    oc, om = ox.map_reads_packed(sub, offs, threads=16)
    gc = counts[sel].cpu().numpy().astype(np.int32)
    gm = matches[sel].cpu().numpy().view(om.dtype).reshape(-1, 2)
    assert (gc == oc).all(), "count mismatch at %s" % sel.cpu().numpy()[np.nonzero(gc != oc)[0][:5]]
    assert (gm[oc > 0, 0] == om[oc > 0, 0]).all() and (gm[oc == 2, 1] == om[oc == 2, 1]).all()
    assert int((oc == 2).sum()) > 1000

    def digest(c, m):
        mm = m.view(-1, 8).to(torch.int64)
        valid1 = (c >= 1).to(torch.int64)[:, None]
        valid2 = (c == 2).to(torch.int64)[:, None]
        w = torch.tensor([3, 5, 7, 11], dtype=torch.int64, device=c.device)
        return c.to(torch.int64) * 1000003 + ((mm[:, :4] * w) * valid1).sum(1) + ((mm[:, 4:] * w * 13) * valid2).sum(1)
    d1 = digest(counts, matches)
    c2, m2 = ix.map_reads_device(rb.bases, rb.offsets, L)
    torch.cuda.synchronize()
    assert torch.equal(digest(c2, m2), d1)                      # determinism
    rev_bases = reads2d.flip(0).contiguous().view(-1)
    c3, m3 = ix.map_reads_device(rev_bases, rb.offsets, L)
    torch.cuda.synchronize()
    assert torch.equal(digest(c3, m3).flip(0), d1)              # order independence
    hits, n_hits = ix.compact_hits_device(counts, matches, n, read_id_base=7, cap=hit_idx.numel() + 10)
    torch.cuda.synchronize()
    h = hits_to_numpy(hits[: int(n_hits.item())])
    assert int(n_hits.item()) == hit_idx.numel()
    assert (h["read_id"] == hit_idx.cpu().numpy() + 7).all()
    assert (h["n"] == counts[hit_idx].cpu().numpy()).all()
    ix.close()
    return info


@pytest.mark.gpu
def test_config2_full_size_cancer_shaped_index(gpu_device, oracle):
    """BASELINE configs[2]'s index at its real size (IDX-C scale 1.0: 136 genes, 15.1 Mbp, ~29 M keys,
    a 0.5 GB table that lives in HBM, the 2.2-bits-per-key filter asked half by half) with 20 M PANEL
    reads: every read with segments and 100 K others re-mapped by the oracle, determinism, order
    independence, ordered compaction."""
    info = _full_size_properties(oracle, "IDX-C", 20_000_000, seed=20240117)
    assert info["n_keys"] > 25_000_000 and info["table_bytes"] > 400_000_000


@pytest.mark.gpu
def test_config2_full_batch(gpu_device, oracle):
    """BASELINE configs[2] at its real BATCH as well: 100 M PANEL pairs = 200 M reads = 30 GB of bases in
    one call — byte offsets past 2^31 and 2^34, the 8192-block plan, a 13 GB workspace — against the cancer-shaped
    index.  Every read with segments and 100 K others re-mapped by the oracle; determinism; order independence;
    ordered compaction.  (The box has 288 GB of HBM; this takes about 90 of them.)"""
    info = _full_size_properties(oracle, "IDX-C", 200_000_000, seed=20240117 + 1000, min_hits=20_000, max_hits=4_000_000,
                                 pairs=True)
    assert info["n_keys"] > 25_000_000


@pytest.mark.gpu
def test_config3_shards_concatenate_to_the_one_shot_list(gpu_device):
    """BASELINE configs[3] in one process: the 8 shards of a batch, each mapped on its own with
    read_id_base = its first read, concatenated in rank order == the hit list of the whole batch
    (values and order).  Shards are slices of the same device arrays (offsets are absolute)."""
    from genefuserust_amd import Indexer, synth
    genes = synth.make_geneset("IDX-D", scale=0.25)
    ix = Indexer.from_gene_slices(genes.seqs, genes.reversed_flags)
    ix.make_index()
    n, L = 4_000_003, 150
    rb = synth.make_reads(genes, n, read_len=L, mix="PANEL", seed=31, device="cuda")
    counts, matches = ix.map_reads_device(rb.bases, rb.offsets, L)
    hits, n_hits = ix.compact_hits_device(counts, matches, n, cap=n // 8)
    want = hits[: int(n_hits.item())].clone()
    assert want.shape[0] > 2000
    for world in (8, 3):
        parts = []
        for r in range(world):
            lo, hi = shard_range(n, r, world)
            c, m = ix.map_reads_device(rb.bases, rb.offsets[lo:hi + 1], L)
            h, k = ix.compact_hits_device(c, m, hi - lo, read_id_base=lo, cap=(hi - lo) // 8)
            parts.append(h[: int(k.item())].clone())
        got = torch.cat(parts)
        assert torch.equal(got, want), world
    ix.close()


def _gpu_shard_worker(rank, world, port, n, L, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from genefuserust_amd import Indexer, synth
        torch.cuda.set_device(0)
        genes = synth.make_geneset("IDX-T", scale=0.2)
        ix = Indexer.from_gene_slices(genes.seqs, genes.reversed_flags, device=0)   # replicated: every rank builds it
        ix.make_index()
        rb = synth.make_reads(genes, n, read_len=L, mix="PANEL", seed=77, device="cuda")  # the same global batch
        lo, hi = shard_range(n, rank, world)
        c, m = ix.map_reads_device(rb.bases, rb.offsets[lo:hi + 1], L)
        h, k = ix.compact_hits_device(c, m, hi - lo, read_id_base=lo, cap=(hi - lo) // 4)
        merged = allgather_hits(h, k)    # cuda tensors over gloo: staged through the host
        q.put((rank, merged.cpu().numpy().copy()))
        if rank == 0:
            c, m = ix.map_reads_device(rb.bases, rb.offsets, L)
            h, k = ix.compact_hits_device(c, m, n, cap=n // 4)
            q.put((-1, h[: int(k.item())].cpu().numpy().copy()))
        ix.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_config3_real_shard_lists_through_the_exchange(gpu_device):
    """The same with two real processes (gloo; both use the one GPU of the box): each rank maps its
    shard with the HIP kernels and the merged list on every rank equals rank 0's one-shot list."""
    world, n, L = 2, 600_001, 150
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_shard_worker, args=(r, world, port, n, L, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in range(world + 1))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got[-1].shape[0] > 1000
    for r in range(world):
        assert np.array_equal(got[r], got[-1])


@pytest.mark.gpu
def test_config4_multi_csv_rebuild_loop(gpu_device, oracle):
    """BASELINE configs[4] on one GPU: reads resident in HBM, 16 CSVs alternating cancer- and
    druggable-shaped gene sets (scaled so that the oracle rebuilds 16 indexes in seconds), the
    index rebuilt per CSV; per CSV the index statistics and every read of a 60 K sample equal the
    oracle's, and the hit list is the compaction of the dense result.  Then the rank plans: 8 and
    32 'ranks' run one after the other reproduce the single-rank lists."""
    from genefuserust_amd import synth
    from genefuserust_amd.indexer import hits_to_numpy
    from genefuserust_amd.multi_csv import scan_multi_csv
    n_csv, L = 16, 150
    sets = [synth.make_geneset("IDX-C" if k % 2 == 0 else "IDX-D", scale=0.02 if k % 2 == 0 else 0.05, seed=1000 + 37 * k)
            for k in range(n_csv)]
    half = 150_000
    synth.MIXES["TEST"] = (0.2, 0.5, 0.3)   # junction-heavy: every CSV that owns reads has thousands of hits
    parts = [synth.make_reads(sets[k], half, read_len=L, mix="TEST", seed=9 + k, device="cuda") for k in (0, 1, 5)]
    bases = torch.cat([p.bases for p in parts])
    n = bases.numel() // L
    offsets = torch.arange(n + 1, device="cuda", dtype=torch.int64) * L
    genesets = [(s.seqs, s.reversed_flags) for s in sets]
    stats = {}
    one = scan_multi_csv(genesets, bases, offsets, L, on_index=lambda k, ix: stats.__setitem__(k, ix.info()))
    assert sorted(one) == list(range(n_csv))
    hb, ho = bases.cpu().numpy(), offsets.cpu().numpy()
    ns = 60_000
    sample = np.concatenate([np.arange(0, ns // 3), np.arange(half, half + ns // 3), np.arange(2 * half, 2 * half + ns // 3)])
    sub = hb.reshape(n, L)[sample].reshape(-1)
    so = np.arange(sample.size + 1, dtype=np.int64) * L
    for k in range(n_csv):
        ox = oracle.OracleIndexer(sets[k].seqs)
        st = ox.stats()
        assert (stats[k]["n_keys"], stats[k]["n_high_keys"], stats[k]["n_unique"]) == (st["n_keys"], st["n_high_keys"], st["m_unique_pos"]), k
        oc, om = ox.map_reads_packed(sub, so, threads=16)
        want = _hits_from_dense(oc, om, 0).numpy()
        want[:, 0] = sample[want[:, 0]]                    # sample index -> read id
        got = one[k].cpu().numpy()
        got = got[np.isin(got[:, 0], sample)]
        assert np.array_equal(got, want), k
        ox.close()
    assert one[0].shape[0] > 1000 and one[1].shape[0] > 1000 and one[5].shape[0] > 1000   # the reads' own gene sets
    # rank plans, run in turn on the one GPU: CSVs >= ranks (no collective) ...
    merged = {}
    for r in range(8):
        merged.update(scan_multi_csv(genesets, bases, offsets, L, rank=r, world=8))
    assert sorted(merged) == list(range(n_csv))
    for k in range(n_csv):
        assert torch.equal(merged[k], one[k]), k
    # ... and fewer CSVs than ranks: the group's shards concatenate to the CSV's list
    few = genesets[:2]
    for k in range(2):
        shards = []
        for r in range(8):
            for j in plan_multi_csv(2, n, r, 8):
                if j.csv == k:
                    shards.append((j.lo, j.hi))
        assert len(shards) == 4
        from genefuserust_amd import Indexer
        ix = Indexer.from_gene_slices(*few[k])
        ix.make_index()
        got = []
        for lo, hi in sorted(shards):
            c, m = ix.map_reads_device(bases, offsets[lo:hi + 1], L)
            h, cnt = ix.compact_hits_device(c, m, hi - lo, read_id_base=lo, cap=hi - lo)
            got.append(h[: int(cnt.item())].clone())
        assert torch.equal(torch.cat(got), one[k]), k
        ix.close()


@pytest.mark.gpu
def test_span_split_of_large_batches(gpu_device, monkeypatch):
    """gf_map_reads_device maps batches beyond GF_SPAN_MAX reads span by span (the kernels keep read
    indices in 32 bits; the workspace is bounded by the span): with the span forced down to 1000
    reads, a ragged 5 K-read batch gives the same dense result."""
    import subprocess, sys, json
    code = r'''
import json, sys, numpy as np, torch
from genefuserust_amd import Indexer, synth
genes = synth.make_geneset("IDX-T", scale=0.05)
ix = Indexer.from_gene_slices(genes.seqs, genes.reversed_flags); ix.make_index()
synth.MIXES["TEST"] = (0.2, 0.5, 0.3)
rb = synth.make_reads(genes, 5003, read_len=150, mix="TEST", seed=3, device="cuda")
c, m = ix.map_reads_device(rb.bases, rb.offsets, 150); torch.cuda.synchronize()
valid = (torch.arange(2, device="cuda")[None, :] < c[:, None].to(torch.int64))[:, :, None]
d = (m.to(torch.int64) * valid).sum().item()
print(json.dumps({"hits": int((c > 0).sum()), "counts": int(c.to(torch.int64).sum()), "digest": int(d)}))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for span in (None, "1000", "64"):
        env = dict(os.environ, PYTHONPATH=root)
        env.pop("GF_SPAN_MAX", None)
        if span:
            env["GF_SPAN_MAX"] = span
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert outs[0]["hits"] > 500
    assert outs[0] == outs[1] == outs[2]
