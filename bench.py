#!/usr/bin/env python
"""Headline benchmark: 150-bp reads/s through Indexer::map_read on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config {1,2,3,4}]

One "step" = one pass of the hot path over one batch of synthetic reads already resident in
HBM: the mapping kernels (K2/K3), the ordered hit compaction (K4) and, for N > 1, the
all-gather of the per-rank hit lists (the path's only exchange).  --config picks the workload
(numbers = index into BASELINE.json `configs`):

  1 (default)  10 M synthetic 150-bp pairs (20 M reads) per GPU vs the druggable-shaped index
               (IDX-D).  N > 1: every rank holds its own 20 M-read shard — weak scaling.
  2            100 M pairs (200 M reads) vs the cancer-shaped index (IDX-C) on one GPU
               (N > 1: the same 200 M reads sharded — strong scaling).
  3            100 M pairs vs IDX-D, read-sharded over the N ranks (25 M reads per rank at
               N = 8), one all-gather of the hit lists per step — strong scaling.
  4            multi-CSV mode: 50 M resident reads, 16 fusion CSVs alternating IDX-C / IDX-D;
               a step rebuilds the index for every CSV a rank owns and maps the reads against it
               (genefuserust_amd/multi_csv.py: CSV k -> rank k % N, no collective when there are
               at least N CSVs).  value = (reads x CSVs) / s, index rebuilds inside the timed region.

One process per GPU.  `python bench.py --gpus N` without WORLD_SIZE in the environment starts its N ranks
itself (a child `python -m torch.distributed.run ... bench.py <same arguments>`, before this process touches the
GPU) and relays rank 0's line and the child's exit code; under torch.distributed.run it is a rank.  Rank 0 prints
ONE JSON line ("Measurement").
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

CONFIGS = {
    # shape, pairs (per rank when weak, total when strong), scaling, seed index (20240115 + config_idx)
    1: dict(shape="IDX-D", pairs=10_000_000, scaling="weak", name="BASELINE configs[1]"),
    2: dict(shape="IDX-C", pairs=100_000_000, scaling="strong", name="BASELINE configs[2]"),
    3: dict(shape="IDX-D", pairs=100_000_000, scaling="strong", name="BASELINE configs[3]"),
    4: dict(shape="IDX-C/IDX-D x16", pairs=25_000_000, scaling="strong", name="BASELINE configs[4]"),
}
SHAPE_TEXT = {"IDX-D": "druggable.hg38-shaped: first 32 gene spans of testdata/cancer.csv",
              "IDX-C": "cancer.hg38-shaped: all 136 gene spans of testdata/cancer.csv",
              "IDX-T": "the 4 gene spans of testdata/fusions.csv"}
READS_TEXT = {
    "pairs": "read pairs per SURVEY.md 8(d): fragments N(300,30) clipped to [150,500] (40 % background / 59.9 % one gene / "
             "0.1 % across a junction for PANEL), R1 = the fragment's first bases, R2 = the reverse complement of its far "
             "end, interleaved R1,R2,R1,R2 as a pair-end scan presents them; value counts reads",
    "independent": "reads are independent draws (40 % background / 59.9 % single-gene / 0.1 % junction for PANEL), half of "
                   "them reverse-complemented — mate-like orientation, not N(300,30) fragment pairs",
}


def _lib_compact_ws_bytes(n: int) -> int:
    from genefuserust_amd import _lib
    return _lib.lib().gf_compact_workspace_bytes(n)


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


def kernel_source_sha() -> str:
    """What the kernels and their launch geometry are made of: sha256 over genefuserust_amd/csrc/* and include/gfmatch.h
    (sorted by name).  tools/make_traffic.py stamps every profiles/hbm_traffic.json entry with it; a line whose loaded
    sources differ from the entry's says `traffic_stale: true`."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "genefuserust_amd", "csrc")
    files = sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith((".h", ".hip")) or f == "Makefile")
    files.append(os.path.join(ROOT, "include", "gfmatch.h"))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def library_sha() -> str:
    import hashlib
    try:
        return hashlib.sha256(open(os.path.join(ROOT, "genefuserust_amd", "libgfmatch.so"), "rb").read()).hexdigest()[:16]
    except Exception:
        return "absent"


def traffic_entry(shape: str, n: int, L: int):
    """Counter-measured fabric bytes per launch for this workload (profiles/hbm_traffic.json), or None."""
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        tj = json.load(open(tpath))
        e = tj.get("%s_%d_%d" % (shape, n, L))
        if e is None:   # the same workload measured at another batch size: per-read traffic is the same
            for k, v in tj.items():
                m = k.split("_")
                if len(m) == 3 and m[0] == shape and m[2] == str(L) and m[1].isdigit():
                    e = dict(v)
                    e["hbm_bytes_per_launch"] = int(v["hbm_bytes_per_launch"] * (n / int(m[1])))
                    e["source"] = "%s (measured at %s reads per launch, scaled by the read count)" % (v.get("source"), m[1])
                    break
        return e
    except Exception:
        return None


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` run plainly: start the N ranks as a CHILD torch.distributed.run (this process has not
    initialised the GPU and never will), let rank 0's JSON line through on our stdout, return the child's exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, default=1, choices=[1, 2, 3, 4], help="BASELINE.json configs[] index")
    ap.add_argument("--pairs", type=int, default=None, help="override: read pairs (per rank for config 1, total otherwise)")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--shape", default=None, choices=["IDX-T", "IDX-D", "IDX-C"], help="override the config's gene set")
    ap.add_argument("--mix", default="PANEL", choices=["PANEL", "WGS"])
    ap.add_argument("--scale", type=float, default=1.0, help="scale of the synthetic gene set (experiments; 1.0 = the named shape)")
    ap.add_argument("--repeat-frac", type=float, default=None, help="fraction of every gene overwritten by repeat-family copies (default 0.02)")
    ap.add_argument("--low-complexity", type=float, default=0.0, help="fraction of every gene overwritten by poly-A / tandem-repeat stretches")
    ap.add_argument("--n-csv", type=int, default=16, help="config 4: fusion CSVs in the list")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline sample time")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-h2d", action="store_true", help="skip the PCIe-inclusive measurement")
    ap.add_argument("--no-packed", action="store_true", help="config 4: map the ASCII reads per CSV instead of packing them once")
    ap.add_argument("--no-pack-sweep", action="store_true", help="skip the boundary sweep (packs x host threads through the C ABI)")
    ap.add_argument("--no-stress", action="store_true", help="skip the stress rows (repeat-rich genes, outside the timed region)")
    ap.add_argument("--reads", default="pairs", choices=["pairs", "independent"],
                    help="pairs (default): synth.make_pair_reads, SURVEY.md 8(d); independent: synth.make_reads (r01/r02's workload)")
    ap.add_argument("--profile-mode", action="store_true", help="only warm-up + timed passes (for rocprofv3 runs): no CPU baseline, "
                    "parity, h2d, packed or per-stage extras")
    ap.add_argument("--calib", action="store_true", help="after the timed steps, launch the counter-calibration kernels of "
                    "tools/gf_calib.hip on the batch's own bases (known byte counts in the same rocprofv3 pass)")
    ap.add_argument("--variant", type=int, default=0, help="first pass: 0 flat pipeline (default), 1 wave-per-read probe-all, 2 wave-per-read seed+verify")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))   # (nothing has touched the GPU yet)
    if args.profile_mode:
        args.no_cpu_baseline = args.no_parity = args.no_h2d = args.no_pack_sweep = args.no_stress = True
    cfg = dict(CONFIGS[args.config])
    if args.shape:
        cfg["shape"] = args.shape
    if args.pairs:
        cfg["pairs"] = args.pairs
    # defaults: about a second of timed region (46 ms inside a 22 s run is more than a driver's sampler of
    # GPU activity can see), a few warm-up steps
    if args.steps is None:
        args.steps = {1: 500, 2: 30, 3: 50, 4: 6}[args.config]
    if args.warmup is None:
        args.warmup = {1: 5, 2: 2, 3: 2, 4: 1}[args.config]

    import numpy as np
    import torch
    import torch.distributed as dist

    from genefuserust_amd import Indexer, synth
    from genefuserust_amd.dist import allgather_hits, shard_range

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d, or plainly "
                         "(bench.py then starts its ranks itself)" % (args.gpus, world, args.gpus))
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
            # drain this rank's own streams first: the hit exchange runs on a communicator and a stream of its own, and
            # two communicators' collectives must not be in flight in different orders on different ranks
            torch.cuda.synchronize()
            dist.barrier()
        torch.cuda.synchronize()

    L = args.read_len
    gene_kw = {}
    if args.repeat_frac is not None:
        gene_kw["repeat_frac"] = args.repeat_frac
    if args.low_complexity:
        gene_kw["low_complexity_frac"] = args.low_complexity
    seed = 20240115 + args.config
    if args.config == 4:
        return bench_multi_csv(args, cfg, world, rank, local_rank, dev, barrier, seed, gene_kw, rehearsal)

    # ---- inputs: index (replicated: every rank builds it) + this rank's shard of reads ----
    total_reads = 2 * cfg["pairs"] * (world if cfg["scaling"] == "weak" else 1)
    lo, hi = shard_range(total_reads, rank, world)   # contiguous shard of the global batch
    n = hi - lo
    genes = synth.make_geneset(cfg["shape"], scale=args.scale, **gene_kw)
    t0 = time.time()
    ix = Indexer.from_gene_slices(genes.seqs, genes.reversed_flags, device=local_rank)
    ix.make_index()
    torch.cuda.synchronize()
    t_index = time.time() - t0
    ix.set_map_variant(args.variant)
    info = ix.info()
    if args.reads == "pairs" and n % 2 == 0:
        reads = synth.make_pair_reads(genes, n // 2, read_len=L, mix=args.mix, seed=seed + 1000 * rank, device=str(dev))
    else:
        args.reads = "independent"
        reads = synth.make_reads(genes, n, read_len=L, mix=args.mix, seed=seed + 1000 * rank, device=str(dev))
    counts = torch.empty(n, dtype=torch.uint8, device=dev)
    matches = torch.empty((n, 2, 4), dtype=torch.int32, device=dev)
    read_id_base = lo
    stream = torch.cuda.current_stream(dev)

    # N > 1: the per-rank hit lists are merged by ONE all-gather per step through the C ABI
    # (gf_allgather_hits_device, include/gfmatch.h: RCCL called by libgfmatch.so; genefuserust_amd/dist.py::RcclHitExchange
    # drives it on a side stream): fixed-capacity blocks, no host round trip, so the exchange of step k overlaps the
    # mapping of step k+1; every step's merged list is finished (waited for, packed on the device) inside the timed region.
    pending = []
    # the step's outputs are preallocated (two sets, alternating: the exchange of step k may still read set k
    # while step k+1 writes the other): nothing is allocated inside the timed region
    cws = int(_lib_compact_ws_bytes(n))
    out_sets = [(torch.empty((max(n // 16, 1), 6), dtype=torch.int64, device=dev), torch.zeros(1, dtype=torch.int64, device=dev),
                 torch.empty(cws, dtype=torch.uint8, device=dev)) for _ in range(2)]
    step_no = [0]
    exch = None
    ranks_seen = [dist.get_world_size() if world > 1 else 1]   # N > 1 over RCCL: what gf_comm_rank reports (below)
    exchange_name = "none (single GPU)"
    if world > 1:
        if rehearsal or os.environ.get("GF_BENCH_PLAIN_ALLGATHER") == "1":
            exchange_name = "allgather_hits over torch.distributed (counts, then padded records; host round trip)"
        else:
            from genefuserust_amd.dist import RcclHitExchange
            # capacity: twice the largest per-rank list of a first pass (the same on every rank); a rank over it is
            # reported by the exchange and fails the run — never cut silently
            ix.map_reads_device(reads.bases, reads.offsets, L, counts, matches)
            _, nh = ix.compact_hits_device(counts, matches, n, read_id_base=read_id_base, cap=n // 16, out=out_sets[0])
            mx = nh.clone()
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            cap_x = min(max(4096, 2 * int(mx.item())), max(n // 16, 1))
            # A communicator that cannot be made, or cannot move data, fails the run on EVERY rank (RcclHitExchange agrees
            # over the torch group after each step that can fail on one rank alone): no switch to another path inside a
            # process that has touched the GPU.  GF_BENCH_PLAIN_ALLGATHER=1 is the explicit opt-out.
            # ... and one that never comes back (a bootstrap that cannot reach its peers) ends the run with a message
            # after GF_BENCH_RCCL_INIT_TIMEOUT_S seconds (default 300) instead of sitting there until the caller's clock.
            import threading
            init_done = threading.Event()
            limit_s = float(os.environ.get("GF_BENCH_RCCL_INIT_TIMEOUT_S", "300"))

            def _watch():
                if not init_done.wait(limit_s):
                    print("rank %d: the RCCL communicator / first exchange through the C ABI did not finish within %.0f s; "
                          "rerun with GF_BENCH_PLAIN_ALLGATHER=1 to take torch.distributed's all-gather instead"
                          % (rank, limit_s), file=sys.stderr, flush=True)
                    os._exit(4)
            threading.Thread(target=_watch, daemon=True).start()
            try:
                exch = RcclHitExchange(cap=cap_x, device=dev)
                exch.first_exchange(out_sets[0][0], nh)
                init_done.set()
            except Exception as e:  # noqa: BLE001
                init_done.set()
                print("rank %d: RCCL exchange through the C ABI failed (%s: %s); rerun with GF_BENCH_PLAIN_ALLGATHER=1 to "
                      "take torch.distributed's all-gather instead" % (rank, type(e).__name__, e), file=sys.stderr, flush=True)
                os._exit(3)   # (a peer's collective may still be waiting on the device: no destructors)
            ranks_seen[0] = exch.comm_world()[1]
            exchange_name = ("gf_allgather_hits_device: one ncclAllGather of %d-record blocks through the C ABI, on a side "
                             "stream, pipelined one step deep" % exch.cap)

    def step(ev=None):
        if ev is not None:
            ev[0].record(stream)
        ix.map_reads_device(reads.bases, reads.offsets, L, counts, matches)
        if ev is not None:
            ev[1].record(stream)
        step_no[0] += 1
        hits, n_hits = ix.compact_hits_device(counts, matches, n, read_id_base=read_id_base, cap=n // 16,
                                              out=out_sets[step_no[0] & 1])
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

    for _ in range(args.warmup):   # (a failing exchange fails the run: no silent fall-back to a slower path)
        step()
    drain()
    barrier()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    out = None
    dbg = os.environ.get("GF_BENCH_DEBUG") == "1"
    for k in range(args.steps):
        out = step(evs[k])
        if dbg:
            print("step %d queued at %.3f ms" % (k, 1e3 * (time.perf_counter() - t0)), file=sys.stderr, flush=True)
    last = drain()
    out = last if last is not None else out
    barrier()
    elapsed = time.perf_counter() - t0
    if dbg:
        print("all steps done at %.3f ms" % (1e3 * elapsed), file=sys.stderr, flush=True)
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    kern_ms = [a.elapsed_time(b) for a, b in evs]
    kern_ms_avg = sum(kern_ms) / len(kern_ms)
    if args.calib and rank == 0:   # outside the timed region: known-byte-count kernels for the PMC passes
        import ctypes
        cal = ctypes.CDLL(os.path.join(ROOT, "tools", "libgfcalib.so"))
        cal.gf_calib_stream.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        sink = torch.zeros(4, dtype=torch.int32, device=dev)
        nb = int(reads.bases.numel()) & ~63
        for mode in (0, 1, 2, 3):
            rc = cal.gf_calib_stream(reads.bases.data_ptr(), nb, mode, sink.data_ptr(), stream.cuda_stream)
            assert rc == 0, "gf_calib_stream mode %d: hip error %d" % (mode, rc)
        torch.cuda.synchronize()
        print("calib: 4 launches over %d bytes" % nb, file=sys.stderr)
    # per-kernel durations of the flat pipeline (HIP events inside the library, launch stream),
    # taken on extra steps outside the timed region
    stage_ms = None
    if args.variant == 0 and not args.profile_mode:
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

    # the packed hand-over (gf_map_reads_packed_device), outside the timed region: the same reads in the
    # kernels' own 2-bit form — what a device-side producer or a host that maps a read set repeatedly hands over
    packed_ms = None
    if args.variant == 0 and rank == 0 and not args.profile_mode:
        t_ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        t_ev[0].record(stream)
        pk, iv = ix.pack_bases_device(reads.bases)
        t_ev[1].record(stream)
        reps = 5
        c2 = torch.empty_like(counts)
        m2 = torch.empty_like(matches)
        ix.map_reads_packed_device(pk, iv, reads.offsets, L, c2, m2)
        t_ev[1].record(stream)
        for _ in range(reps):
            ix.map_reads_packed_device(pk, iv, reads.offsets, L, c2, m2)
        t_ev[2].record(stream)
        torch.cuda.synchronize()
        same = bool(torch.equal(c2, counts))
        packed_ms = {"map_ms": t_ev[1].elapsed_time(t_ev[2]) / reps, "identical_counts": same,
                     "bytes_per_base": 0.375}
        del pk, iv, c2, m2
    # the fixed-length hand-over (gf_map_reads_fixed_device), outside the timed region: the same reads without the
    # int64 offsets array — every read of the batch has L bases, the kernels compute where it starts
    fixed_ms = None
    if args.variant == 0 and rank == 0 and not args.profile_mode and L <= 320:
        c3 = torch.empty_like(counts)
        m3 = torch.empty_like(matches)
        ix.map_reads_fixed_device(reads.bases, L, c3, m3)
        same = bool(torch.equal(c3, counts))

        def timed5(fn):
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(5):
                fn()
            e1.record(stream)
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / 5
        t_off, t_fix = [], []
        for _ in range(3):   # alternating, so that neither form gets the warmer caches
            t_off.append(timed5(lambda: ix.map_reads_device(reads.bases, reads.offsets, L, c3, m3)))
            t_fix.append(timed5(lambda: ix.map_reads_fixed_device(reads.bases, L, c3, m3)))
        fixed_ms = {"map_ms": sum(t_fix) / 3, "offsets_form_map_ms_same_loop": sum(t_off) / 3,
                    "identical_counts": same, "bytes_per_read_not_loaded": 8}
        fixed_ms["kernel_reads_per_s"] = n / (fixed_ms["map_ms"] * 1e-3)
        del c3, m3
    value = total_reads * args.steps / elapsed
    if world == 1:
        n_hits_total = int(out[1].item())
    elif isinstance(out, tuple):  # RcclHitExchange: (merged, totals = [records, overflow flag, per-rank counts ..])
        tot = out[1].cpu().tolist()
        if tot[1]:
            raise SystemExit("hit exchange over capacity (cap %d): %s" % (exch.cap, tot[2:]))
        n_hits_total = int(tot[0])
    else:
        n_hits_total = int(out.shape[0])

    result = {
        "metric": "150bp_reads_per_s_map_read_vs_%s_shaped_index" % {"IDX-D": "druggable", "IDX-C": "cancer", "IDX-T": "testdata"}[cfg["shape"]],
        "value": value,
        "unit": "reads/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": cfg["scaling"],
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic" if not rehearsal else "synthetic (REHEARSAL on one GPU over gloo: not a measurement)",
        "config": {
            "workload": "%s: %d synthetic %d-bp read pairs (%d reads) %s vs %s (%s, %d bp), mix %s; %s"
                        % (cfg["name"] if not (args.shape or args.pairs) else cfg["name"] + " (overridden)",
                           total_reads // 2, L, total_reads,
                           "in total, %d reads on each of %d ranks" % (n, world) if world > 1 else "on one GPU",
                           cfg["shape"], SHAPE_TEXT[cfg["shape"]], info["total_bp"], args.mix, READS_TEXT[args.reads]),
            "baseline_config": args.config,
            "reads_per_gpu_per_step": n,
            "read_len": L,
            "index_shape": cfg["shape"],
            "index_keys": info["n_keys"],
            "index_table_bytes": info["table_bytes"],
            "index_build_s": round(t_index, 3),
            "hits_per_step": n_hits_total,
            "exchange": exchange_name,
            "ranks_seen": ranks_seen[0],
            "parallelism": "reads sharded over %d rank(s), index replicated, all-gather of hit records" % world
                           if world > 1 else "single GPU",
        },
    }
    if gene_kw:
        result["config"]["gene_synthesis"] = gene_kw

    # PCIe-inclusive rate, every rank through its own link at the same time (never `value`): what N GPUs buy a HOST —
    # the device-resident rate of one GPU is already 20x what one link feeds, the links are what scales
    h2d = None
    if not args.no_h2d:
        barrier()
        try:
            h2d = h2d_inclusive(ix, reads, n, L)
        except Exception as e:  # noqa: BLE001 — a reported extra, never the value
            h2d = {"error": "%s: %s" % (type(e).__name__, e)}
        if world > 1:
            v = torch.tensor([h2d.get("reads_per_s", 0.0), h2d.get("host_GBps", 0.0)], dtype=torch.float64,
                             device="cpu" if rehearsal else dev)
            allv = [torch.zeros_like(v) for _ in range(world)]
            dist.all_gather(allv, v)
            per = [[float(x) for x in t.cpu()] for t in allv]
            h2d = dict(h2d, per_rank_reads_per_s=[p[0] for p in per], reads_per_s_all_ranks=sum(p[0] for p in per),
                       host_GBps_all_ranks=sum(p[1] for p in per),
                       note="every rank streams its own 8 M-read sample from pinned host memory through its own link, concurrently")
    if rank == 0:
        result["roofline"] = roofline_object(args, cfg, n, L, kern_ms_avg, stage_ms)
        if h2d is not None:
            result["h2d_inclusive"] = h2d
        if packed_ms:
            packed_ms["kernel_reads_per_s"] = n / (packed_ms["map_ms"] * 1e-3)
            result["packed_input"] = packed_ms
        if fixed_ms:
            result["fixed_length_input"] = fixed_ms
        if world == 1 and not args.no_stress and args.config == 1 and not (args.shape or args.repeat_frac is not None or args.low_complexity):
            try:
                result["stress"] = stress_rows(args, cfg, local_rank, dev, L, seed)
            except Exception as e:  # noqa: BLE001 — reported extras, never the value
                result["stress"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not args.no_pack_sweep:
            try:
                result["pack_sweep"] = pack_sweep(genes, reads, n, L)
            except Exception as e:  # noqa: BLE001 — a reported extra, never the value
                result["pack_sweep"] = {"error": "%s: %s" % (type(e).__name__, e)}
        # ---- parity spot check + CPU baseline (oracle = CPU restatement; never on the product path) ----
        want_cpu = world == 1 and not args.no_cpu_baseline
        want_parity = not args.no_parity
        if want_cpu or want_parity:
            from oracle import oracle_py
            ox = oracle_py.OracleIndexer(genes.seqs)
            cores = usable_cores()
            if want_parity:
                # every read of this rank's step that came back with segments, plus 100 K reads drawn at random, re-mapped
                # by the oracle (what tests/test_gpu_parity.py::test_full_size_properties does)
                hit_ids = ((counts > 0) & (counts < 255)).nonzero().flatten()
                g = torch.Generator(device="cpu")
                g.manual_seed(1234)
                sample = torch.randint(0, n, (min(n, 100_000),), generator=g).to(dev)
                ids = torch.unique(torch.cat([hit_ids, sample]))   # (sorted)
                ns = int(ids.numel())
                b = reads.bases.view(n, L)[ids].reshape(-1).cpu().numpy()
                o = np.arange(ns + 1, dtype=np.int64) * L
                oc, om = ox.map_reads_packed(b, o, threads=cores)
                gc = counts[ids].cpu().numpy().astype(np.int32)
                gm = matches[ids].cpu().numpy().view(om.dtype).reshape(ns, 2)
                ok = bool((gc == oc).all())
                nz = oc > 0
                ok = ok and bool((gm[nz, 0] == om[nz, 0]).all()) and bool((gm[oc == 2, 1] == om[oc == 2, 1]).all())
                result["parity"] = {"checked_reads": ns, "bit_exact": ok, "reads_with_segments": int(nz.sum()),
                                    "gpu_reads_with_segments": int(hit_ids.numel()),
                                    "what": "every read of the step with segments + 100 000 reads drawn at random, re-mapped by the oracle"}
            if want_cpu:
                result["cpu_baseline"] = cpu_baseline(ox, reads, n, L, cores, args.cpu_seconds)
        print(json.dumps(result), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


COMPULSORY_BYTES_PER_READ = lambda L: L + 8 + 1   # its bases, its int64 offset, its count byte: what any design must move


def roofline_object(args, cfg, n, L, kern_ms_avg, stage_ms):
    """`frac` = bytes the L2s exchanged with the fabric per pass (rocprofv3 PMC, calibrated: profiles/hbm_traffic.json,
    tools/make_traffic.py) / summed duration of the pass's kernels / 8 TB/s — what the memory system delivers.
    `algorithmic_frac` is the contract's formula (706 B per 150-bp read, 544 B of them 'one 8-byte index
    slot per probe' that this design never fetches) over the same time; `compulsory_frac` charges only a read's own
    bytes.  When no counter measurement exists for the workload, `frac` falls back to the algorithmic figure and says so."""
    secs = kern_ms_avg * 1e-3
    algo = algo_bytes_per_read(L) * n  # bytes per launch
    algo_gbps = algo / secs / 1e9
    comp_gbps = COMPULSORY_BYTES_PER_READ(L) * n / secs / 1e9
    # the counter entry is of the default workload only: PANEL pair reads, default gene synthesis, flat pipeline
    default_workload = (args.scale == 1.0 and args.mix == "PANEL" and args.reads == "pairs" and args.variant == 0
                        and args.repeat_frac is None and not args.low_complexity)
    te = traffic_entry(cfg["shape"], n, L) if default_workload else None
    traffic = te["hbm_bytes_per_launch"] if te else None
    src_sha = kernel_source_sha()
    stale = bool(te) and te.get("kernel_src_sha") != src_sha
    measured = traffic / secs / 1e9 if traffic else None
    achieved = measured if measured is not None else algo_gbps
    return {
        "bound": "hbm",
        "binding_resource": "vector instruction issue (measured, r04): nine more vector instructions per filter look-up, no memory "
                            "access, cost seed+verify +14 % of its time; 786 M vector + 157 M scalar instructions per 20 M reads, the "
                            "vector ALUs issuing 73 % of the cycles at four waves per SIMD.  No MFMA: integer hashing.  The fraction "
                            "below is of the HBM peak because that is the roofline a byte-moving path is held against; r03's "
                            "'4.8 ps per L2 hit + 14.8 ps per miss' fitted because hits and instructions both scale with the look-ups "
                            "(DESIGN.md 5)",
        "kernel": {0: "gf_map_reads_device = 4 kernels: gf_k_seedverify_stream (dominant) + gf_k_probe_filter + "
                      "gf_k_probe_buckets + gf_k_map_reads_list; their summed duration",
                   1: "gf_k_map_reads_short<4,0> (wave per read, probe-all)",
                   2: "gf_k_map_reads_short<4,1> (wave per read, seed+verify)"}[args.variant],
        "stage_ms": stage_ms,
        "achieved": achieved,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS,
        "frac_definition": ("measured: L2 <-> fabric bytes (incl. Infinity-Cache hits) = (32 B x TCC_EA0_RDREQ_DRAM_32B + WRITE_SIZE) "
                            "per pass / kernel time / 8 TB/s"
                            if measured is not None else
                            "ALGORITHMIC (no counter measurement for this workload): 706 B per read / kernel time / 8 TB/s"),
        "traffic": traffic,
        "traffic_source": (te or {}).get("source"),
        "traffic_build": None if not te else {"git_head": te.get("git_head"), "kernel_src_sha": te.get("kernel_src_sha"),
                                              "lib_sha": te.get("lib_sha")},
        "traffic_stale": stale if te else None,
        "loaded_build": {"kernel_src_sha": src_sha, "lib_sha": library_sha()},
        "hbm_only": None,
        "hbm_only_note": "not separable: rocprofv3 on gfx950 exposes the CPC CPF GRBM SPI SQ TA TCA TCC TCP TD blocks only "
                         "(counter_defs.yaml) - no Infinity-Cache (MALL) or memory-controller counter; TCC_EA0_RDREQ_DRAM* counts "
                         "requests 'destined for DRAM' whether the Infinity Cache or HBM serves them",
        "traffic_read_bytes": (te or {}).get("read_bytes_per_launch"),
        "traffic_write_bytes": (te or {}).get("write_bytes_per_launch"),
        "algorithmic_GBps": algo_gbps,
        "algorithmic_frac": algo_gbps / HBM_PEAK_GBS,
        "algorithmic_bytes_per_read": algo_bytes_per_read(L),
        "compulsory_GBps": comp_gbps,
        "compulsory_frac": comp_gbps / HBM_PEAK_GBS,
        "compulsory_bytes_per_read": COMPULSORY_BYTES_PER_READ(L),
        "reads_per_launch": n,
        "kernel_ms_avg": kern_ms_avg,
        "kernel_reads_per_s": n / secs,
    }


def cpu_baseline(ox, reads, n, L, cores, cpu_seconds):
    # bounded sample: time a probe, then size the sample for ~cpu_seconds of CPU work
    probe = min(n, 200_000)
    b = reads.bases[: probe * L].cpu().numpy()
    o = reads.offsets[: probe + 1].cpu().numpy()
    t1 = time.perf_counter()
    ox.map_reads_packed(b, o, threads=cores)
    rate = probe / (time.perf_counter() - t1)
    ns = int(min(n, max(probe, rate * cpu_seconds)))
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
    return {
        "value": ns / dt,
        "unit": "reads/s",
        "cores": cores,
        "kind": "port",
        "sample": "first %d reads of the same batch, oracle/indexer_oracle.cc (CPU restatement of the "
                  "reference algorithm: 2^32-bit bitmap + hash map + ordered vote map, one read per call, "
                  "%d threads pulling 1000-read packs); %.1f s" % (ns, cores, dt),
        "value_4_threads": t4,
    }


def stress_rows(args, cfg, local_rank, dev, L, seed):
    """The headline's pass on gene sets and read mixes that are harder than SURVEY.md 8(d)'s defaults (2 % repeats, PANEL),
    outside the timed region, 8 M reads each: human introns are 40-50 % interspersed repeats, the default genes 2 %.
    Kernel-only rates like `roofline.kernel_reads_per_s`; the full-size rows are profiles/r04_stress_*.json."""
    import torch
    from genefuserust_amd import Indexer, synth
    pairs = 4_000_000
    rows = {}
    specs = [("repeat10", dict(shape=cfg["shape"], gene_kw=dict(repeat_frac=0.10), mix="PANEL")),
             ("repeat30", dict(shape=cfg["shape"], gene_kw=dict(repeat_frac=0.30), mix="PANEL")),
             ("lowcomplexity5", dict(shape=cfg["shape"], gene_kw=dict(low_complexity_frac=0.05), mix="PANEL")),
             ("wgs_mix", dict(shape=cfg["shape"], gene_kw={}, mix="WGS")),
             ("idx_t", dict(shape="IDX-T", gene_kw={}, mix="PANEL"))]
    stream = torch.cuda.current_stream(dev)
    for name, sp in specs:
        genes = synth.make_geneset(sp["shape"], **sp["gene_kw"])
        ix = Indexer.from_gene_slices(genes.seqs, genes.reversed_flags, device=local_rank)
        ix.make_index()
        reads = synth.make_pair_reads(genes, pairs, read_len=L, mix=sp["mix"], seed=seed + 77, device=str(dev))
        n = 2 * pairs
        counts = torch.empty(n, dtype=torch.uint8, device=dev)
        matches = torch.empty((n, 2, 4), dtype=torch.int32, device=dev)
        ix.map_reads_device(reads.bases, reads.offsets, L, counts, matches)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3
        e0.record(stream)
        for _ in range(reps):
            ix.map_reads_device(reads.bases, reads.offsets, L, counts, matches)
        e1.record(stream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        info = ix.info()
        rows[name + "_reads_per_s"] = n / (ms * 1e-3)
        rows[name] = {"map_ms": round(ms, 4), "reads": n, "index_keys": info["n_keys"], "high_keys": info["n_high_keys"],
                      "dupe_keys": info["n_dupe_keys"], "reads_with_segments": int(((counts > 0) & (counts < 255)).sum().item()),
                      "shape": sp["shape"], "mix": sp["mix"], "gene_synthesis": sp["gene_kw"]}
        ix.close()
        del reads, counts, matches, genes
    rows["what"] = ("kernel-only reads/s of gf_map_reads_device on 8 M reads (4 M PANEL/WGS pairs), outside the timed region; "
                    "repeatNN = NN % of every gene overwritten by copies of a 20-element family of 300-bp repeats "
                    "(headline: 2 %), lowcomplexity5 = 5 % poly-A / tandem-repeat stretches")
    return rows


def pack_sweep(genes, reads, n, L):
    """The boundary at the reference's granularity (never `value`): packs of P pairs x T host threads through the C ABI —
    gf_map_reads_hits and gf_stream_submit / collect, pageable and pinned sources — by tools/pack_sweep.cpp in a child
    process on the first 8 M reads of the batch.  The reference: PACK_SIZE = 1000 pairs (common.rs:23), one
    Indexer::map_read per read from t-1 consumer threads (pescanner.rs:296-311, 374-518)."""
    from tools.bench_pack_sweep import run_sweep
    ns = min(n, 8_000_000)
    j = run_sweep(genes.seqs, reads.bases[: ns * L].cpu().numpy(), reads.offsets[: ns + 1].cpu().numpy(), seconds=0.2,
                  extra=["--packs", "125,250,500,1000,4000,16000,64000,256000,1000000"])
    cells = j["cells"]
    threads = sorted({c["threads"] for c in cells})
    packs = sorted({c["pack_pairs"] for c in cells})
    tab = {}
    for c in cells:
        key = "%s, %s" % (c["entry"], c["mem"])
        tab.setdefault(key, {p: [None] * len(threads) for p in packs})[c["pack_pairs"]][threads.index(c["threads"])] = \
            round(c["reads_per_s"] / 1e6, 1)
    smallest = {}
    for key, rows in tab.items():
        if 8 in threads:
            ok = [p for p in packs if (rows[p][threads.index(8)] or 0) >= 50.0]
            smallest[key] = ok[0] if ok else None
    return {"unit": "M reads/s", "threads": threads, "pack_pairs": packs,
            "rows": {k: {str(p): v for p, v in rows.items()} for k, rows in tab.items()},
            "smallest_pack_pairs_with_8_threads_over_50M_reads_per_s": smallest,
            "us_per_single_read_call": j["us_per_single_read_call"], "seconds_per_cell": j["seconds_per_cell"],
            "reads": ns,
            "what": "tools/pack_sweep.cpp through the C ABI only: T threads on one index, each mapping packs of P pairs (2 P reads); "
                    "calls of up to 8192 reads take the zero-copy route (one launch), larger ones the batch route"}


def h2d_inclusive(ix, reads, n, L):
    """PCIe-inclusive rate (SURVEY.md §8d asks for it beside the kernel-only figure; never `value`): the
    same reads starting in pinned host memory, through the double-buffered streaming entry
    (gf_stream_*: copy of pack k+1 overlaps the kernels of pack k), hit records back on the host."""
    import numpy as np
    import torch
    from genefuserust_amd.stream import MapStream, pinned_empty
    ns = min(n, 8_000_000)
    pack = 1_000_000
    hb = pinned_empty(ns * L, np.uint8)
    ho = pinned_empty(ns + 1, np.int64)
    hb[:] = reads.bases[: ns * L].cpu().numpy()
    ho[:] = reads.offsets[: ns + 1].cpu().numpy()
    out = {}
    with MapStream(ix, max_reads=pack, max_bytes=pack * L + 64, depth=3) as ms:
        for rep in range(2):  # the first pass warms the arenas
            t0 = time.perf_counter()
            total = 0
            inflight = 0
            for p0 in range(0, ns, pack):
                p1 = min(ns, p0 + pack)
                if inflight == ms.depth:
                    total += ms.collect().shape[0]
                    inflight -= 1
                ms.submit(hb, ho[p0:p1 + 1], read_id_base=p0)
                inflight += 1
            while inflight:
                total += ms.collect().shape[0]
                inflight -= 1
            dt = time.perf_counter() - t0
        out = {"reads_per_s": ns / dt, "host_GBps": (ns * L + 8 * (ns + 1)) / dt / 1e9, "reads": ns, "pack_reads": pack,
               "hits": int(total), "bytes_per_read": L + 8,
               "entry": "gf_stream_submit/collect, pinned host buffers, depth 3"}
    # the same packs handed over in packed form (gf_pack_bases_host + gf_stream_submit_packed): 6 bytes per 16 bases over
    # the link.  Two figures: the stream alone from pre-packed pinned memory (what a host that keeps its reads packed
    # gets — multi-CSV mode, a FASTQ parser that emits the form), and the host packer's own rate on this box's cores.
    try:
        from genefuserust_amd.stream import pack_bases_host
        cores = usable_cores()
        t0 = time.perf_counter()
        pk, iv = pack_bases_host(hb, threads=cores, pinned=True)
        t_first = time.perf_counter() - t0          # (includes the first touch of the pinned output)
        from genefuserust_amd import _lib as _gl
        t0 = time.perf_counter()
        _gl.check(_gl.lib().gf_pack_bases_host(hb.ctypes.data, hb.size, pk.ctypes.data, iv.ctypes.data, cores))
        t_pack = time.perf_counter() - t0
        with MapStream(ix, max_reads=pack, max_bytes=pack * L + 64, depth=3) as ms:
            for rep in range(2):
                t0 = time.perf_counter()
                total_p = 0
                inflight = 0
                for p0 in range(0, ns, pack):
                    p1 = min(ns, p0 + pack)
                    if inflight == ms.depth:
                        total_p += ms.collect().shape[0]
                        inflight -= 1
                    ms.submit_packed(pk, iv, ho[p0:p1 + 1], read_id_base=p0)
                    inflight += 1
                while inflight:
                    total_p += ms.collect().shape[0]
                    inflight -= 1
                dtp = time.perf_counter() - t0
        out["packed"] = {"reads_per_s_prepacked": ns / dtp, "host_GBps_prepacked": (ns * L * 0.375 + 8 * (ns + 1)) / dtp / 1e9,
                         "bytes_per_read": L * 0.375 + 8, "hits": int(total_p), "same_hits_as_ascii": int(total_p) == int(total),
                         "host_pack_GBps_of_ascii": hb.size / t_pack / 1e9, "host_pack_threads": cores,
                         "host_pack_reads_per_s": ns / t_pack, "host_pack_first_call_s": round(t_first, 3),
                         "entry": "gf_pack_bases_host (AVX2) + gf_stream_submit_packed, pinned, depth 3"}
        del pk, iv
    except Exception as e:  # noqa: BLE001 — a reported extra
        out["packed"] = {"error": "%s: %s" % (type(e).__name__, e)}
    # the link alone: one large pinned H2D copy
    d = torch.empty(ns * L, dtype=torch.uint8, device=reads.bases.device)
    t = torch.from_numpy(hb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    d.copy_(t, non_blocking=True)
    torch.cuda.synchronize()
    out["link_GBps_one_copy"] = ns * L / (time.perf_counter() - t0) / 1e9
    return out


def bench_multi_csv(args, cfg, world, rank, local_rank, dev, barrier, seed, gene_kw, rehearsal):
    """BASELINE configs[4]: index rebuilt per CSV over a resident read set (fusion_scan.rs:62-188)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from genefuserust_amd import Indexer, synth
    from genefuserust_amd.multi_csv import plan_multi_csv
    from genefuserust_amd.dist import allgather_hits
    L = args.read_len
    n = 2 * cfg["pairs"]
    n_csv = args.n_csv
    shapes = ["IDX-C" if k % 2 == 0 else "IDX-D" for k in range(n_csv)]
    # CSV k = its shape's gene spans filled from its own seed: 16 different indexes, two sizes
    jobs = plan_multi_csv(n_csv, n, rank, world)
    sets = {j.csv: synth.make_geneset(shapes[j.csv], scale=args.scale, seed=1000 + 37 * j.csv, **gene_kw) for j in jobs}
    # the resident reads: drawn from CSV 0's and CSV 1's genes (every rank holds the same set)
    base_sets = [synth.make_geneset(("IDX-C", "IDX-D")[k], scale=args.scale, seed=1000 + 37 * k, **gene_kw) for k in (0, 1)]
    half = n // 2
    if args.reads == "pairs" and half % 2 == 0 and (n - half) % 2 == 0:
        ra = synth.make_pair_reads(base_sets[0], half // 2, read_len=L, mix=args.mix, seed=seed, device=str(dev))
        rb = synth.make_pair_reads(base_sets[1], (n - half) // 2, read_len=L, mix=args.mix, seed=seed + 1, device=str(dev))
    else:
        args.reads = "independent"
        ra = synth.make_reads(base_sets[0], half, read_len=L, mix=args.mix, seed=seed, device=str(dev))
        rb = synth.make_reads(base_sets[1], n - half, read_len=L, mix=args.mix, seed=seed + 1, device=str(dev))
    bases = torch.cat([ra.bases, rb.bases])
    offsets = torch.arange(n + 1, device=dev, dtype=torch.int64) * L
    del ra, rb
    groups = {}
    if world > 1 and n_csv < world:  # shared CSVs: one process group per CSV (created collectively, in order)
        inner = world // n_csv
        for k in range(n_csv):
            g = tuple(range(k * inner, (k + 1) * inner))
            groups[g] = dist.new_group(list(g))
    build_ms, map_ms, hit_counts, last = [], [], {}, {}
    pack_ms = []
    # parity sample of one IDX-C-shaped and one IDX-D-shaped CSV: device buffers made before the timed region
    keep_for = [] if args.no_parity or rank != 0 else [j.csv for j in jobs[:2]]
    kept, n_keys_of = {}, {}
    for j in jobs[:2]:
        if j.csv in keep_for:
            ns = min(j.hi - j.lo, 200_000)
            kept[j.csv] = (torch.empty(ns, dtype=torch.uint8, device=dev), torch.empty((ns, 2, 4), dtype=torch.int32, device=dev))

    def step(record):
        packed = None   # once per step: the reads' 2-bit form, shared by all the CSVs of the step
        for j in jobs:
            gs = sets[j.csv]
            t0 = time.perf_counter()
            ix = Indexer.from_gene_slices(gs.seqs, gs.reversed_flags, device=local_rank)
            ix.make_index()
            t1 = time.perf_counter()
            m = j.hi - j.lo
            if packed is None and not args.no_packed:
                packed = ix.pack_bases_device(bases)
                torch.cuda.synchronize()
                if record:
                    pack_ms.append(1e3 * (time.perf_counter() - t1))
                t1 = time.perf_counter()
            if packed is not None:
                counts, matches = ix.map_reads_packed_device(packed[0], packed[1], offsets[j.lo:j.hi + 1], L)
            else:
                counts, matches = ix.map_reads_device(bases, offsets[j.lo:j.hi + 1], L)
            hits, n_hits = ix.compact_hits_device(counts, matches, m, read_id_base=j.lo, cap=max(m // 8, 4096))
            if len(j.group) > 1:
                merged = allgather_hits(hits, n_hits, group=groups[j.group])
                k = merged.shape[0]
            else:
                k = int(n_hits.item())   # (waits for the launches)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            if record:
                build_ms.append(1e3 * (t1 - t0))
                map_ms.append(1e3 * (t2 - t1))
                hit_counts[j.csv] = k
                if j.csv in keep_for:   # the parity sample stays on the device; it is downloaded after the clock stops
                    ns = min(m, 200_000)
                    kept[j.csv][0].copy_(counts[:ns])
                    kept[j.csv][1].copy_(matches[:ns])
                    n_keys_of[j.csv] = ix.info()["n_keys"]
            ix.close()

    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    for c, (kc, km) in kept.items():
        last[c] = (n_keys_of[c], kc.cpu().numpy().copy(), km.cpu().numpy().copy())
    value = n * n_csv * args.steps / elapsed
    result = {
        "metric": "150bp_reads_x_csvs_per_s_multi_csv_mode_index_rebuilt_per_csv",
        "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic" if not rehearsal else "synthetic (REHEARSAL on one GPU over gloo: not a measurement)",
        "config": {
            "workload": "%s: %d resident %d-bp reads (%d pairs), %d fusion CSVs alternating IDX-C / IDX-D gene spans (own "
                        "seed each), index rebuilt per CSV inside the timed region; value counts a read once per CSV; %s"
                        % (cfg["name"], n, L, n // 2, n_csv, READS_TEXT[args.reads]),
            "baseline_config": 4, "reads_resident": n, "n_csv": n_csv, "read_len": L,
            "csvs_of_rank0": [j.csv for j in jobs],
            "reads_form": "ASCII" if args.no_packed else "packed once per step (gf_pack_bases_device), mapped per CSV with gf_map_reads_packed_device",
            "pack_ms_rank0": [round(x, 2) for x in pack_ms[-1:]],
            "index_build_ms_rank0": [round(x, 2) for x in build_ms[-len(jobs):]],
            "map_ms_rank0": [round(x, 2) for x in map_ms[-len(jobs):]],
            "hits_per_csv_rank0": hit_counts,
            "parallelism": ("CSV k -> rank k %% %d, every rank maps all reads against its CSVs, no collective" % world)
                           if n_csv >= world else
                           ("%d ranks per CSV, reads sharded inside the group, one all-gather per CSV" % (world // n_csv)),
        },
    }
    if rank == 0:
        tot_build, tot_map = sum(build_ms), sum(map_ms)
        algo_gbps = algo_bytes_per_read(L) * n * len(jobs) * args.steps / (tot_map * 1e-3) / 1e9 if tot_map else None
        # counter-measured bytes of one whole step (every kernel: rebuilds, packing, mapping passes, compaction),
        # profiles/hbm_traffic.json key config4_<reads>x<csvs>_<L>, over the step's wall time
        te = None
        try:
            te = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json"))).get("config4_%dx%d_%d" % (n, n_csv, L))
        except Exception:
            pass
        step_s = elapsed / args.steps
        traffic = te["hbm_bytes_per_launch"] if te and world == 1 and not args.no_packed else None
        measured = traffic / step_s / 1e9 if traffic else None
        result["roofline"] = {
            "bound": "hbm", "kernel": "one step = per CSV: index rebuild (K1) + mapping pass (4 kernels) + compaction; the reads packed once",
            "achieved": measured if measured is not None else algo_gbps,
            "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": (measured if measured is not None else algo_gbps) / HBM_PEAK_GBS if (measured or algo_gbps) else None,
            "frac_definition": ("measured: L2 <-> fabric bytes (incl. Infinity-Cache hits) = (32 B x TCC_EA0_RDREQ_DRAM_32B + WRITE_SIZE) of every "
                                "kernel of a step / the step's wall time / 8 TB/s"
                                if measured is not None else
                                "ALGORITHMIC (no counter measurement for this workload): 706 B per read and CSV / host-timed mapping time / 8 TB/s"),
            "traffic": traffic, "traffic_source": (te or {}).get("source"),
            "traffic_build": None if not te else {"git_head": te.get("git_head"), "kernel_src_sha": te.get("kernel_src_sha"),
                                                  "lib_sha": te.get("lib_sha")},
            "traffic_stale": (te.get("kernel_src_sha") != kernel_source_sha()) if te else None,
            "loaded_build": {"kernel_src_sha": kernel_source_sha(), "lib_sha": library_sha()},
            "hbm_only": None,
            "hbm_only_note": "not separable: rocprofv3 on gfx950 exposes no Infinity-Cache (MALL) or memory-controller counter",
            "algorithmic_GBps": algo_gbps, "algorithmic_frac": algo_gbps / HBM_PEAK_GBS if algo_gbps else None,
            "share_of_step_in_index_rebuild": tot_build / (tot_build + tot_map) if tot_build + tot_map else None,
        }
        if not args.no_parity:
            from oracle import oracle_py
            cores = usable_cores()
            checked = {}
            for j in jobs[:2]:  # one IDX-C-shaped and one IDX-D-shaped CSV
                ox = oracle_py.OracleIndexer(sets[j.csv].seqs)
                nk, gc, gm = last[j.csv]
                ns = gc.shape[0]
                b = bases[j.lo * L:(j.lo + ns) * L].cpu().numpy()
                o = np.arange(ns + 1, dtype=np.int64) * L
                oc, om = ox.map_reads_packed(b, o, threads=cores)
                gmv = gm.view(om.dtype).reshape(ns, 2)
                nz = oc > 0
                ok = bool((gc.astype(np.int32) == oc).all()) and bool((gmv[nz, 0] == om[nz, 0]).all()) and \
                    bool((gmv[oc == 2, 1] == om[oc == 2, 1]).all()) and nk == ox.stats()["n_keys"]
                checked[j.csv] = {"checked_reads": ns, "bit_exact": ok, "reads_with_segments": int(nz.sum())}
                del ox
            result["parity"] = checked
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
