"""The two independent restatements (C++ oracle, Python model) must agree, and the
C++ oracle must reproduce the committed golden vectors."""
import json
import os

import numpy as np
import pytest

from oracle import indexer_model as M
from tests.helpers import ACGT, branch_genes, branch_reads, rand_seq, rc

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "branch_cases.json")


def test_golden_inputs_match_builders():
    g = json.load(open(GOLDEN))
    genes, rev = branch_genes()
    assert g["genes"] == [None if x is None else x.decode() for x in genes]
    assert g["reversed"] == rev
    reads = branch_reads(genes)
    assert [c["label"] for c in g["cases"]] == [l for l, _ in reads]
    assert [c["read"] for c in g["cases"]] == [r.decode() for _, r in reads]


def test_oracle_reproduces_golden(oracle):
    g = json.load(open(GOLDEN))
    genes = [None if x is None else x.encode() for x in g["genes"]]
    ox = oracle.OracleIndexer(genes)
    assert ox.stats() == g["stats"]
    for c in g["cases"]:
        assert ox.map_read(c["read"].encode()) == [tuple(m) for m in c["expect"]], c["label"]
    for k, n, sites in g["index"][::7]:
        got_n, got_sites = ox.lookup(k)
        assert got_n == n and got_sites == [tuple(s) for s in sites]
    # batch + threads entry point returns the same as one-by-one
    from genefuserust_amd.synth import ragged_batch
    from tests.helpers import matches_to_tuples
    bases, offsets = ragged_batch([c["read"].encode() for c in g["cases"]])
    for threads in (1, 3):
        counts, matches = ox.map_reads_packed(bases, offsets, threads=threads)
        assert matches_to_tuples(counts, matches) == [[tuple(m) for m in c["expect"]] for c in g["cases"]]


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_random_repeat_rich_agreement(oracle, seed):
    """Small alphabet-poor genes (many 2..5x and >=6x k-mers, palindromes) and reads
    stitched from them: index content and map_read must agree between the models."""
    rng = np.random.default_rng(100 + seed)
    unit = rand_seq(rng, 40)
    genes = []
    for gi in range(3):
        g = bytearray(rand_seq(rng, 700 + 100 * gi))
        for _ in range(int(rng.integers(2, 9))):
            p = int(rng.integers(0, len(g) - 40))
            g[p:p + 40] = unit
        if seed == 1:
            g[100:140] = b"AT" * 20          # low-complexity, self-overlapping, palindromic
        if seed == 2:
            g[50] = ord("N"); g[300:310] = bytes(g[300:310]).lower()
        genes.append(bytes(g))
    ox = oracle.OracleIndexer(genes)
    mx = M.IndexModel([g.decode() for g in genes])
    keys = set(int(k) for k in ox.keys())
    assert keys == set(mx.table.keys())
    for k in list(keys)[::5]:
        n, sites = ox.lookup(k)
        v = mx.table[k]
        assert (n == -2 and v is M.HIGH) or sorted(v) == sites
    up = [g.upper() for g in genes]
    for t in range(60):
        a, b = rng.integers(0, 3, size=2)
        L = int(rng.choice([120, 150, 180]))
        brk = int(rng.integers(25, L - 25))
        pa = int(rng.integers(brk, len(up[a]) - 1))
        pb = int(rng.integers(0, len(up[b]) - (L - brk)))
        read = up[a][pa - brk + 1:pa + 1] + up[b][pb:pb + L - brk]
        if t % 3 == 0:
            read = rc(read)
        if t % 5 == 0:
            rr = bytearray(read); rr[int(rng.integers(0, L))] = ACGT[rng.integers(0, 4)]; read = bytes(rr)
        assert ox.map_read(read) == mx.map_read(read.decode()), (seed, t)
