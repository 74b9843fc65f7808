#!/usr/bin/env python3
"""Differential fuzzing of the mapping path on the GPU box: random gene sets (repeat-rich),
ragged batches mixing every read-length class, N-rich and lower-case reads, junction-heavy
mixes — every read's SeqMatch list against the CPU oracle.  Exits non-zero on the first
difference and prints the offending read.

    python tools/fuzz_parity.py [rounds] [seed]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from genefuserust_amd import Indexer, synth  # noqa: E402
from oracle import oracle_py  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    total = 0
    for rd in range(rounds):
        rng = np.random.default_rng(seed0 * 1000 + rd)
        shape = ["IDX-T", "IDX-D", "IDX-C"][rd % 3]
        scale = float(rng.choice([0.002, 0.005, 0.02])) * (0.3 if shape == "IDX-C" else 1.0)
        # GF_FUZZ_REPEAT: fraction of every gene overwritten by repeat-family copies (default 0.02; 0.3 makes HIGH and
        # 2..5-fold seeds common: the vote bound's treatment of in-table windows that cannot vote)
        rep = float(os.environ.get("GF_FUZZ_REPEAT", "0.02"))
        genes = synth.make_geneset(shape, scale=scale, seed=1000 + rd, repeat_frac=rep)
        ix = Indexer.from_gene_slices(genes.seqs, genes.reversed_flags)
        ix.make_index()
        ix.set_map_variant(0)
        ox = oracle_py.OracleIndexer(genes.seqs)
        synth.MIXES["FUZZ"] = tuple(rng.dirichlet([1.0, 2.0, 1.0]))
        parts = []
        for L in rng.choice([60, 64, 100, 150, 151, 160, 161, 200, 256, 257, 270, 272, 300, 320, 321, 500, 1100],
                            size=6, replace=False):
            L = int(L)
            usable = [len(s) for s in genes.seqs if len(s) >= 2 * L]
            if not usable:
                continue
            rb = synth.make_reads(genes, int(rng.integers(2000, 9000)), read_len=L, mix="FUZZ", seed=int(rng.integers(1 << 30)))
            b = rb.bases.numpy().reshape(-1, L).copy()
            # damage: N runs, lower case, truncations (ragged lengths)
            for _ in range(b.shape[0] // 20):
                r, p = int(rng.integers(b.shape[0])), int(rng.integers(L))
                b[r, p:p + int(rng.integers(1, 4))] = ord("N")
            for _ in range(b.shape[0] // 50):
                r, p = int(rng.integers(b.shape[0])), int(rng.integers(L))
                b[r, p:p + int(rng.integers(1, 30))] |= 0x20
            lens = np.full(b.shape[0], L)
            cut = rng.random(b.shape[0]) < 0.2
            lens[cut] = rng.integers(0, L + 1, size=int(cut.sum()))
            parts += [bytes(b[i, :lens[i]]) for i in range(b.shape[0])]
        if not parts:   # (a tiny gene set and six long read lengths: nothing to map this round)
            ix.close()
            continue
        order = rng.permutation(len(parts))
        reads = [parts[i] for i in order]
        bases, offsets = synth.ragged_batch(reads)
        pad = int(rng.integers(0, 16))   # misaligned start
        bases = np.concatenate([np.frombuffer(b"G" * pad, dtype=np.uint8), bases])
        offsets = offsets + pad
        mx = int(rng.choice([int(np.diff(offsets).max()), 160, 256, 320, 4096]))
        d_b, d_o = torch.from_numpy(bases).cuda(), torch.from_numpy(offsets).cuda()
        counts, matches = ix.map_reads_device(d_b, d_o, mx)
        torch.cuda.synchronize()
        c = counts.cpu().numpy()[:len(reads)].astype(np.int32)
        m = matches.cpu().numpy().view(oracle_py.ORC_SEQMATCH).reshape(-1, 2)[:len(reads)]
        oc, om = ox.map_reads_packed(bases, offsets, threads=8)
        lens = np.diff(offsets)
        too_long = lens > mx
        assert (c[too_long] == 255).all(), "too-long marking"
        ok = ~too_long
        bad = np.nonzero(ok & (c != oc))[0]
        if bad.size == 0:
            two = ok & (oc >= 1)
            bad = np.nonzero(two & (m[:, 0] != om[:, 0]))[0]
            if bad.size == 0:
                bad = np.nonzero(ok & (oc == 2) & (m[:, 1] != om[:, 1]))[0]
        if bad.size:
            r = int(bad[0])
            print("MISMATCH round", rd, "shape", shape, "scale", scale, "read", r, "len", int(lens[r]), "max_read_len", mx)
            print(" device", int(c[r]), m[r], "\n oracle", int(oc[r]), om[r])
            print(" read", reads[r][:400])
            return 1
        total += len(reads)
        print("round %d ok: %s scale %.4f, %d reads (%d with segments), max_read_len %d, pad %d" %
              (rd, shape, scale, len(reads), int((oc > 0).sum()), mx, pad), flush=True)
        ix.close()
    print("fuzz ok:", total, "reads")
    return 0


if __name__ == "__main__":
    sys.exit(main())
