#!/usr/bin/env python3
"""Measure the front of the paired-end path on one MI355X, everything
resident in HBM: FASTQ text of R1 and R2 -> records (gf_fastq_index_device +
gf_fastq_gather_device) -> fast_merge (find + write) -> mapping of the merged reads.

Synthetic pairs per SURVEY.md §8(d) (fragment N(300,30) clipped to [150,500], 150-bp reads,
0.5 % errors, 80 % of them low quality); records of 317 bytes ("@r%010d/1", bases, "+",
qualities).  One JSON line: per-stage ms (HIP events on the launch stream), pairs/s, the
stages' algorithmic bytes against the 8 TB/s HBM peak, and a parity sample against the
CPU oracle (records, merged reads)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tools.bench_merge import make_pairs  # noqa: E402


def make_text(bases, quals, n, L, mate, dev):
    """(n, 13 + 2L + 4) bytes: '@r0000000042/1\\n' bases '\\n+\\n' quals '\\n'"""
    name_w = 14
    rec = name_w + 1 + L + 1 + 1 + 1 + L + 1
    t = torch.empty((n, rec), dtype=torch.uint8, device=dev)
    t[:, 0] = ord("@")
    t[:, 1] = ord("r")
    idx = torch.arange(n, device=dev, dtype=torch.int64)
    for d in range(10):
        t[:, 2 + d] = (48 + (idx // (10 ** (9 - d))) % 10).to(torch.uint8)
    t[:, 12] = ord("/")
    t[:, 13] = ord(str(mate))
    t[:, 14] = 10
    t[:, 15:15 + L] = bases.view(n, L)
    t[:, 15 + L] = 10
    t[:, 16 + L] = ord("+")
    t[:, 17 + L] = 10
    t[:, 18 + L:18 + 2 * L] = quals.view(n, L)
    t[:, 18 + 2 * L] = 10
    return t.reshape(-1)


def timed(fn, steps, warmup):
    for _ in range(warmup):
        out = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=5_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--check", type=int, default=5000)
    a = ap.parse_args()
    from genefuserust_amd import Indexer
    from genefuserust_amd.fastq import fastq_cut_device
    from genefuserust_amd.read_pair import fast_merge_device
    from genefuserust_amd.synth import make_geneset
    dev = torch.device("cuda", 0)
    genes = make_geneset("IDX-D")
    ix = Indexer.from_gene_slices(genes.seqs)
    ix.make_index()
    n, L = a.pairs, a.read_len
    (lb, lq, rb, rq), off = make_pairs(n, L, 20240201, dev)
    t1 = make_text(lb, lq, n, L, 1, dev)
    t2 = make_text(rb, rq, n, L, 2, dev)
    del lb, lq, rb, rq
    torch.cuda.synchronize()
    ms_cut1, b1 = timed(lambda: fastq_cut_device(ix, t1), a.steps, a.warmup)
    ms_cut2, b2 = timed(lambda: fastq_cut_device(ix, t2), a.steps, a.warmup)
    assert b1.n_records == b2.n_records == n
    ms_merge, mg = timed(lambda: fast_merge_device(ix, b1.bases, b1.quals, b1.offsets, b2.bases, b2.quals, b2.offsets, L),
                         a.steps, a.warmup)
    mb, mq, moff, mdiff = mg
    mlen = int((moff[1:] - moff[:-1]).max().item())
    ms_map, (cnt, mat) = timed(lambda: ix.map_reads_device(mb, moff, mlen), a.steps, a.warmup)
    n_merged = int(((moff[1:] - moff[:-1]) > 0).sum())
    # algorithmic bytes: text read once, bases + qualities + offsets written once (cut); bases +
    # qualities of both reads read, merged bases + qualities written (merge)
    cut_bytes = t1.numel() + t2.numel() + 2 * (2 * n * L + 8 * n)
    merge_bytes = 4 * n * L + 2 * int(mb.numel())
    # parity sample against the oracle
    from oracle import oracle_py
    k = min(a.check, n)
    rec = t1.numel() // n
    h1, h2 = t1[:k * rec].cpu().numpy().tobytes(), t2[:k * rec].cpu().numpy().tobytes()
    w1, w2 = oracle_py.fastq_cut(h1), oracle_py.fastq_cut(h2)
    o1 = b1.offsets[:k + 1].cpu().numpy()
    g1 = b1.bases[:o1[-1]].cpu().numpy().tobytes()
    q1 = b1.quals[:o1[-1]].cpu().numpy().tobytes()
    bad = sum(g1[o1[i]:o1[i + 1]] != w1[i][1] or q1[o1[i]:o1[i + 1]] != w1[i][3] for i in range(k))
    mo = moff[:k + 1].cpu().numpy()
    hb, hq = mb[:mo[-1]].cpu().numpy().tobytes(), mq[:mo[-1]].cpu().numpy().tobytes()
    dd = mdiff[:k].cpu().numpy()
    for i in range(k):
        w = oracle_py.fast_merge(w1[i][1], w1[i][3], w2[i][1], w2[i][3])
        got = (hb[mo[i]:mo[i + 1]], hq[mo[i]:mo[i + 1]], int(dd[i])) if mo[i + 1] > mo[i] else None
        bad += got != w
    total = ms_cut1 + ms_cut2 + ms_merge
    print(json.dumps({
        "metric": "read pairs per second through FASTQ cutting + fast_merge (device-resident text)",
        "value": n / (total / 1e3), "unit": "pairs/s", "pairs": n, "read_len": L,
        "stage_ms": {"fastq_cut_R1": round(ms_cut1, 3), "fastq_cut_R2": round(ms_cut2, 3), "fast_merge": round(ms_merge, 3),
                     "map_merged_reads": round(ms_map, 3)},
        "merged_fraction": n_merged / n, "text_bytes": int(t1.numel() + t2.numel()),
        "roofline": {"bound": "hbm", "peak": 8000.0, "unit": "GB/s",
                     "fastq_cut": {"algorithmic_bytes": cut_bytes, "achieved": cut_bytes / ((ms_cut1 + ms_cut2) / 1e3) / 1e9,
                                   "frac": cut_bytes / ((ms_cut1 + ms_cut2) / 1e3) / 8e12},
                     "fast_merge": {"algorithmic_bytes": merge_bytes, "achieved": merge_bytes / (ms_merge / 1e3) / 1e9,
                                    "frac": merge_bytes / (ms_merge / 1e3) / 8e12}},
        "parity": {"checked_pairs": k, "mismatches": int(bad)}}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
