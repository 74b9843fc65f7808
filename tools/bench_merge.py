#!/usr/bin/env python3
"""Measure gf_k_fast_merge on synthetic pairs resident in HBM.

Pairs per SURVEY.md §8(d): fragment length N(300,30) clipped to [150,500], R1 = first 150
bases, R2 = reverse complement of the far end; 0.5 % sequencing errors, 80 % of them with a
low quality ('#'..'0'), the others high; qualities otherwise 'A'..'J'.  Prints one JSON line:
pairs/s, the two launches' time (HIP events on the launch stream), algorithmic bytes
(bases + qualities of both reads read once, merged bases + qualities written once) against
the 8 TB/s HBM peak, and a parity check of a sample against the CPU oracle."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def make_pairs(n, L, seed, dev):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    comp = torch.zeros(256, dtype=torch.uint8, device=dev)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    out = [torch.empty((n, L), dtype=torch.uint8, device=dev) for _ in range(4)]
    chunk = 1 << 20
    ar = torch.arange(L, device=dev)
    for c0 in range(0, n, chunk):
        m = min(chunk, n - c0)
        flen = (300 + 30 * torch.randn(m, generator=g, device=dev)).round().clamp_(150, 500).to(torch.int64)
        frag = acgt[torch.randint(0, 4, (m, 500), generator=g, device=dev)]
        r1 = frag[:, :L].clone()
        idx = (flen[:, None] - 1 - ar[None, :])
        r2 = comp[torch.gather(frag, 1, idx).long()]
        for r, q in ((r1, out[1]), (r2, out[3])):
            err = torch.rand((m, L), generator=g, device=dev) < 0.005
            low = torch.rand((m, L), generator=g, device=dev) < 0.8
            r[err] = acgt[torch.randint(0, 4, (int(err.sum()),), generator=g, device=dev)]
            qq = (65 + torch.randint(0, 10, (m, L), generator=g, device=dev)).to(torch.uint8)
            lo = (35 + torch.randint(0, 14, (m, L), generator=g, device=dev)).to(torch.uint8)
            qq = torch.where(err & low, lo, qq)
            q[c0:c0 + m] = qq
        out[0][c0:c0 + m] = r1
        out[2][c0:c0 + m] = r2
    off = torch.arange(0, (n + 1) * L, L, dtype=torch.int64, device=dev)
    return [t.reshape(-1) for t in out], off


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=10_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--check", type=int, default=20000)
    a = ap.parse_args()
    from genefuserust_amd import Indexer
    from genefuserust_amd.read_pair import fast_merge_device
    dev = torch.device("cuda", 0)
    ix = Indexer.from_gene_slices([b"ACGT" * 64])   # the merge only needs the index's device
    ix.make_index()
    (lb, lq, rb, rq), off = make_pairs(a.pairs, a.read_len, 20240201, dev)
    torch.cuda.synchronize()
    for _ in range(a.warmup):
        res = fast_merge_device(ix, lb, lq, off, rb, rq, off, a.read_len)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(a.steps):
        res = fast_merge_device(ix, lb, lq, off, rb, rq, off, a.read_len)
    e1.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / a.steps
    ms = e0.elapsed_time(e1) / a.steps
    bases, quals, moff, diff = res
    n_merged = int(((moff[1:] - moff[:-1]) > 0).sum())
    alg = 4 * a.pairs * a.read_len + 2 * int(bases.numel())
    # parity sample against the oracle
    from oracle import oracle_py
    k = min(a.check, a.pairs)
    L = a.read_len
    h = [t[:k * L].cpu().numpy().tobytes() for t in (lb, lq, rb, rq)]
    mo = moff[:k + 1].cpu().numpy()
    mb, mq = bases[:mo[-1]].cpu().numpy().tobytes(), quals[:mo[-1]].cpu().numpy().tobytes()
    dd = diff[:k].cpu().numpy()
    bad = 0
    for i in range(k):
        w = oracle_py.fast_merge(*(x[i * L:(i + 1) * L] for x in h))
        got = (mb[mo[i]:mo[i + 1]], mq[mo[i]:mo[i + 1]], int(dd[i])) if mo[i + 1] > mo[i] else None
        bad += got != w
    print(json.dumps({"metric": "pairs merged per second (fast_merge, device-resident)", "value": a.pairs / (ms / 1e3),
                      "unit": "pairs/s", "pairs": a.pairs, "read_len": L, "merged_fraction": n_merged / a.pairs,
                      "ms_per_step": ms, "wall_ms_per_step": wall * 1e3, "steps": a.steps,
                      "roofline": {"bound": "hbm", "achieved": alg / (ms / 1e3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                   "frac": alg / (ms / 1e3) / 8e12, "algorithmic_bytes": alg},
                      "parity": {"checked": k, "mismatches": bad}}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
