#!/usr/bin/env python
"""Headline benchmark: 150-bp reads/s through Indexer::map_read on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one pass of the hot path over one batch of synthetic reads already
resident in HBM: the mapping kernel (K2/K3), the ordered hit compaction (K4) and,
for N > 1, the all-gather of the per-rank hit lists (the path's only exchange).
Workload at N = 1 is BASELINE.json configs[1]: 10 M synthetic 150-bp pairs
(20 M reads) against the druggable-shaped index (IDX-D).  For
N > 1 every rank holds its own 20 M-read shard of a global batch (weak scaling),
one process per GPU, launched by torch.distributed.run.

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_READ_150 = 706  # 150 + 8 + 8*68 + 4
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec


def algo_bytes_per_read(L: int) -> int:
    p1 = (L - 16) // 2 + 1 if L >= 16 else 0
    return L + 8 + 8 * p1 + 4


def usable_cores() -> int:
    """Host cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:  # cgroup v2
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        try:  # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            pr = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                cores = max(1, min(cores, q // pr))
        except Exception:
            pass
    return cores


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=10_000_000, help="read pairs per rank per step")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--shape", default="IDX-D", choices=["IDX-T", "IDX-D", "IDX-C"])
    ap.add_argument("--mix", default="PANEL", choices=["PANEL", "WGS"])
    ap.add_argument("--scale", type=float, default=1.0, help="scale of the synthetic gene set (experiments; 1.0 = the named shape)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline sample time")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--variant", type=int, default=0, help="first pass: 0 flat pipeline (default), 1 wave-per-read probe-all, 2 wave-per-read seed+verify")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    from genefuserust_amd import Indexer, synth
    from genefuserust_amd.dist import allgather_hits
    from genefuserust_amd.indexer import hits_to_numpy

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (gfmatch has no CPU fallback)")
    # GF_BENCH_REHEARSAL=1: every rank uses cuda:0 and the collective runs on gloo — only to
    # exercise the N>1 code path on a one-GPU box; never a measurement
    rehearsal = os.environ.get("GF_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- inputs: index (replicated: every rank builds it) + this rank's shard of reads ----
    L = args.read_len
    n = 2 * args.pairs
    genes = synth.make_geneset(args.shape, scale=args.scale)
    t0 = time.time()
    ix = Indexer.from_gene_slices(genes.seqs, genes.reversed_flags, device=local_rank)
    ix.make_index()
    torch.cuda.synchronize()
    t_index = time.time() - t0
    ix.set_map_variant(args.variant)
    info = ix.info()
    reads = synth.make_reads(genes, n, read_len=L, mix=args.mix, seed=20240116 + rank, device=str(dev))
    counts = torch.empty(n, dtype=torch.uint8, device=dev)
    matches = torch.empty((n, 2, 4), dtype=torch.int32, device=dev)
    read_id_base = rank * n
    stream = torch.cuda.current_stream(dev)

    # N > 1: the per-rank hit lists are merged by one asynchronous all-gather per step
    # (genefuserust_amd/dist.py::HitExchange): fixed-capacity blocks, no host round trip, so the
    # exchange of step k overlaps the mapping of step k+1; every step's merged list is finished
    # (waited for and packed on the device) inside the timed region.
    exch = None
    if world > 1:
        if rehearsal or os.environ.get("GF_BENCH_PLAIN_ALLGATHER") == "1":
            exch = None  # gloo on host copies (or asked for): the plain allgather_hits path
        else:
            from genefuserust_amd.dist import HitExchange
            exch = HitExchange(cap=max(4096, n // 512), device=dev)
    pending = []

    def step(ev=None):
        if ev is not None:
            ev[0].record(stream)
        ix.map_reads_device(reads.bases, reads.offsets, L, counts, matches)
        if ev is not None:
            ev[1].record(stream)
        hits, n_hits = ix.compact_hits_device(counts, matches, n, read_id_base=read_id_base, cap=n // 16)
        if world > 1:
            if exch is None:
                return allgather_hits(hits, n_hits)
            pending.append(exch.start(hits, n_hits))
            if len(pending) > 1:  # finish the previous step's exchange while this step's kernels run
                return exch.finish(pending.pop(0))
            return None
        return hits, n_hits

    def drain():
        out = None
        while pending:
            out = exch.finish(pending.pop(0))
        return out

    try:
        for _ in range(args.warmup):
            step()
        drain()
    except Exception as e:  # noqa: BLE001 — the exchange failing the same way on every rank: use the plain path
        if exch is None:
            raise
        print("HitExchange failed (%s): falling back to allgather_hits" % e, file=sys.stderr)
        exch = None
        pending.clear()
        for _ in range(args.warmup):
            step()
    barrier()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    out = None
    for k in range(args.steps):
        out = step(evs[k])
    last = drain()
    out = last if last is not None else out
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    kern_ms = [a.elapsed_time(b) for a, b in evs]
    kern_ms_avg = sum(kern_ms) / len(kern_ms)
    # per-kernel durations of the flat pipeline (HIP events inside the library, launch stream),
    # taken on extra steps outside the timed region
    stage_ms = None
    if args.variant == 0:
        ix.set_profiling(True)
        acc = [0.0, 0.0, 0.0, 0.0]
        reps = 5
        for _ in range(reps):
            ix.map_reads_device(reads.bases, reads.offsets, L, counts, matches)
            ms = ix.last_stage_ms()
            acc = [a + b for a, b in zip(acc, ms)]
        ix.set_profiling(False)
        names = ["gf_k_seedverify_stream", "gf_k_probe_filter", "gf_k_probe_buckets", "gf_k_map_reads_list"]
        stage_ms = {k: round(a / reps, 4) for k, a in zip(names, acc) if k}

    total_reads = n * world * args.steps
    value = total_reads / elapsed
    if world == 1:
        n_hits_total = int(out[1].item())
    elif isinstance(out, tuple):  # HitExchange: (merged, total, overflow)
        assert not bool(out[2].item()), "hit exchange capacity exceeded: raise HitExchange cap"
        n_hits_total = int(out[1].item())
    else:
        n_hits_total = int(out.shape[0])

    result = {
        "metric": "150bp_reads_per_s_map_read_vs_druggable_shaped_index",
        "value": value,
        "unit": "reads/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic" if not rehearsal else "synthetic (REHEARSAL on one GPU over gloo: not a measurement)",
        "config": {
            "workload": "%s: %d synthetic %d-bp read pairs (%d reads) per GPU vs %s (%s, %d bp), mix %s"
                        % ({"IDX-D": "BASELINE configs[1]", "IDX-C": "BASELINE configs[2] shape",
                            "IDX-T": "BASELINE configs[0] gene set"}[args.shape], args.pairs, L, n, args.shape,
                           {"IDX-D": "druggable.hg38-shaped: first 32 gene spans of testdata/cancer.csv",
                            "IDX-C": "cancer.hg38-shaped: all 136 gene spans of testdata/cancer.csv",
                            "IDX-T": "the 4 gene spans of testdata/fusions.csv"}[args.shape],
                           info["total_bp"], args.mix),
            "reads_per_gpu_per_step": n,
            "read_len": L,
            "index_shape": args.shape,
            "index_keys": info["n_keys"],
            "index_table_bytes": info["table_bytes"],
            "index_build_s": round(t_index, 3),
            "hits_per_step": n_hits_total,
            "parallelism": "reads sharded over %d rank(s), index replicated, all-gather of hit records" % world
                           if world > 1 else "single GPU",
        },
    }

    if rank == 0:
        # ---- roofline of the dominant kernel (gf_k_map_reads) ----
        algo = algo_bytes_per_read(L) * n  # bytes per launch
        achieved = algo / (kern_ms_avg * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                key = "%s_%d_%d" % (args.shape, n, L)
                if key in tj:
                    traffic = tj[key]["hbm_bytes_per_launch"]
            except Exception:
                traffic = None
        result["roofline"] = {
            "bound": "hbm",
            "kernel": {0: "gf_map_reads_device = 4 kernels: gf_k_seedverify_stream + gf_k_probe_filter + "
                          "gf_k_probe_buckets + gf_k_map_reads_list; achieved uses their summed duration",
                       1: "gf_k_map_reads_short<4,0> (wave per read, probe-all)",
                       2: "gf_k_map_reads_short<4,1> (wave per read, seed+verify)"}[args.variant],
            "stage_ms": stage_ms,
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "algorithmic_bytes_per_read": algo_bytes_per_read(L),
            "reads_per_launch": n,
            "kernel_ms_avg": kern_ms_avg,
            "kernel_reads_per_s": n / (kern_ms_avg * 1e-3),
        }

        # ---- parity spot check + CPU baseline (oracle = CPU restatement; never on the product path) ----
        want_cpu = world == 1 and not args.no_cpu_baseline
        want_parity = not args.no_parity
        if want_cpu or want_parity:
            from oracle import oracle_py
            ox = oracle_py.OracleIndexer(genes.seqs)
            cores = usable_cores()
            if want_parity:
                ns = min(n, 400_000)
                b = reads.bases[: ns * L].cpu().numpy()
                o = reads.offsets[: ns + 1].cpu().numpy()
                oc, om = ox.map_reads_packed(b, o, threads=cores)
                gc = counts[:ns].cpu().numpy().astype(np.int32)
                gm = matches[:ns].cpu().numpy().view(om.dtype).reshape(ns, 2)
                ok = bool((gc == oc).all())
                nz = oc > 0
                ok = ok and bool((gm[nz, 0] == om[nz, 0]).all()) and bool((gm[oc == 2, 1] == om[oc == 2, 1]).all())
                result["parity"] = {"checked_reads": ns, "bit_exact": ok, "reads_with_segments": int(nz.sum())}
            if want_cpu:
                # bounded sample: time a probe, then size the sample for ~cpu_seconds of CPU work
                probe = min(n, 200_000)
                b = reads.bases[: probe * L].cpu().numpy()
                o = reads.offsets[: probe + 1].cpu().numpy()
                t1 = time.perf_counter()
                ox.map_reads_packed(b, o, threads=cores)
                rate = probe / (time.perf_counter() - t1)
                ns = int(min(n, max(probe, rate * args.cpu_seconds)))
                b = reads.bases[: ns * L].cpu().numpy()
                o = reads.offsets[: ns + 1].cpu().numpy()
                t1 = time.perf_counter()
                ox.map_reads_packed(b, o, threads=cores)
                dt = time.perf_counter() - t1
                t4 = None
                if cores >= 4:
                    ns4 = max(1, ns // max(1, cores // 4))
                    t1 = time.perf_counter()
                    ox.map_reads_packed(b[: ns4 * L], o[: ns4 + 1], threads=4)
                    t4 = ns4 / (time.perf_counter() - t1)
                result["cpu_baseline"] = {
                    "value": ns / dt,
                    "unit": "reads/s",
                    "cores": cores,
                    "kind": "port",
                    "sample": "first %d reads of the same batch, oracle/indexer_oracle.cc (CPU restatement of the "
                              "reference algorithm: 2^32-bit bitmap + hash map + ordered vote map, one read per call, "
                              "%d threads pulling 1000-read packs); %.1f s" % (ns, cores, dt),
                    "value_4_threads": t4,
                }
        print(json.dumps(result), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
