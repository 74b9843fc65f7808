#!/usr/bin/env python3
"""Differential fuzzing of fast_merge and the FASTQ cutter on the GPU box against the oracle.
    python tools/fuzz_merge.py [rounds] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from genefuserust_amd import Indexer  # noqa: E402
from genefuserust_amd.fastq import fastq_cut_device  # noqa: E402
from genefuserust_amd.read_pair import fast_merge_device, pack_reads  # noqa: E402
from oracle import oracle_py  # noqa: E402
from tests.test_fast_merge import make_pairs  # noqa: E402
from tests.test_fastq import synth_fastq  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    ix = Indexer.from_gene_slices([b"ACGT" * 64])
    ix.make_index()
    dev = torch.device("cuda", 0)
    n_pairs = n_rec = 0
    for rd in range(rounds):
        rng = np.random.default_rng(seed0 * 7919 + rd)
        lo = int(rng.choice([30, 60, 100, 140, 150, 200, 240]))
        hi = lo + int(rng.choice([1, 2, 12, 60]))
        pairs = make_pairs(seed0 * 100 + rd, int(rng.integers(1500, 5000)), read_len=(lo, hi))
        hint = int(rng.choice([hi, 160, 256, 100, 301]))
        lb, lo_ = pack_reads([p[0] for p in pairs]); lq, _ = pack_reads([p[1] for p in pairs])
        rb, ro = pack_reads([p[2] for p in pairs]); rq, _ = pack_reads([p[3] for p in pairs])
        t = [torch.from_numpy(a).to(dev) for a in (lb, lq, lo_, rb, rq, ro)]
        bases, quals, off, diff = fast_merge_device(ix, *t, hint)
        torch.cuda.synchronize()
        b, q, o, d = bases.cpu().numpy().tobytes(), quals.cpu().numpy().tobytes(), off.cpu().numpy(), diff.cpu().numpy()
        b2 = fast_merge_device(ix, *t, hint, with_quals=False)[0]   # the pair pipeline's bases-only writer
        torch.cuda.synchronize()
        if b2.cpu().numpy().tobytes() != b:
            print("MERGE MISMATCH round", rd, ": the bases-only writer differs from the byte-per-lane writer")
            return 1
        for i, p in enumerate(pairs):
            w = oracle_py.fast_merge(*p)
            g = (b[o[i]:o[i + 1]], q[o[i]:o[i + 1]], int(d[i])) if o[i + 1] > o[i] else None
            if g != w:
                print("MERGE MISMATCH round", rd, "pair", i, "hint", hint, p, "\n device", g, "\n oracle", w)
                return 1
        n_pairs += len(pairs)
        text = synth_fastq(seed0 * 31 + rd, int(rng.integers(100, 6000)), lens=(int(rng.integers(0, 80)), int(rng.integers(81, 400))),
                           ragged_quality=bool(rd % 3 == 0))
        if rd % 4 == 1:
            text = text[:-int(rng.integers(1, 40))]
        want = oracle_py.fastq_cut(text)
        batch = fastq_cut_device(ix, torch.from_numpy(np.frombuffer(text, dtype=np.uint8).copy()).to(dev))
        torch.cuda.synchronize()
        off = batch.offsets.cpu().numpy()
        bb, qq = batch.bases.cpu().numpy().tobytes(), batch.quals.cpu().numpy().tobytes()
        assert batch.n_records == len(want), (rd, batch.n_records, len(want))
        for i, w in enumerate(want):
            ln = len(w[1])
            eq = w[3][:ln] + b"!" * max(0, ln - len(w[3]))
            if bb[off[i]:off[i + 1]] != w[1] or qq[off[i]:off[i + 1]] != eq:
                print("FASTQ MISMATCH round", rd, "record", i, w)
                return 1
        n_rec += len(want)
        print("round %d ok: %d pairs (reads %d..%d, hint %d), %d records" % (rd, len(pairs), lo, hi - 1, hint, len(want)), flush=True)
    print("fuzz ok:", n_pairs, "pairs,", n_rec, "records")
    return 0


if __name__ == "__main__":
    sys.exit(main())
