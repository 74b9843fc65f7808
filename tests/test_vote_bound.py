"""The vote bound of gf_table.h (gf_vote_bound_pairs), on the CPU.

gf_k_seedverify_stream ends a read with [] when NO diagonal can collect the 20 first-pass votes of the gate
(indexer.rs:353-360).  r03 replaced "fewer than 20 windows left" by a structural bound: two voters of one
diagonal that are at most 7 windows apart have only in-table windows between them.  Three things are checked:

  1. the recurrence the device runs (compiled here from the very header, with g++) equals an exhaustive search
     over voter sets on pair-aligned masks and is an upper bound of it on ragged ones;
  2. the statement itself, against the oracle: for reads over repeat-rich gene sets (short genes, both strands,
     strand junctions, N, 2..5-fold and >= 6-fold keys) every diagonal's actual voters obey it, and count1 never
     exceeds the bound computed from the windows whose key is in the table;
  3. the pattern the kernel relies on: even pairs ruled out -> 12 for a 150-base read.
"""
import ctypes
import itertools
import os
import subprocess

import numpy as np
import pytest

from tests.helpers import branch_genes, branch_reads, rand_seq, rc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SHIM = r"""
#include "gf_table.h"
extern "C" int vote_bound(const uint32_t* x, int npairs) {
  switch (npairs) {
    case 8: return gf_vote_bound_pairs<8>(x);
    case 20: return gf_vote_bound_pairs<20>(x);
    case 34: return gf_vote_bound_pairs<34>(x);
    case 37: return gf_vote_bound_pairs<37>(x);
    case 61: return gf_vote_bound_pairs<61>(x);
    case 77: return gf_vote_bound_pairs<77>(x);
  }
  return -1;
}
"""


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    d = tmp_path_factory.mktemp("vb")
    src, so = str(d / "vb.cc"), str(d / "libvb.so")
    open(src, "w").write(SHIM)
    subprocess.run(["g++", "-std=c++17", "-O1", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "genefuserust_amd", "csrc"),
                    src, "-o", so], check=True)
    L = ctypes.CDLL(so)
    L.vote_bound.argtypes = [ctypes.c_void_p, ctypes.c_int]
    return L


def device_bound(shim, standing_windows, npairs):
    """standing_windows: iterable of 0/1 per stride-2 window -> the header's bound (pairs standing when either window is)."""
    w = list(standing_windows) + [0] * (2 * npairs + 2)
    x = np.zeros((npairs + 15) // 16 + 1, dtype=np.uint32)
    for P in range(npairs):
        if w[2 * P] or w[2 * P + 1]:
            x[P >> 4] |= np.uint32(1 << (2 * (P & 15)))
    return shim.vote_bound(x.ctypes.data, npairs)


def exhaustive_bound(can):
    """largest V within `can` with: u < v in V, v - u <= 7  =>  every window of u..v is in `can`."""
    idx = [i for i, c in enumerate(can) if c]
    best = 0
    for r in range(len(idx), 0, -1):
        if r <= best:
            break
        for V in itertools.combinations(idx, r):
            if all(V[j] - V[i] > 7 or all(can[V[i]:V[j] + 1]) for i in range(r) for j in range(i + 1, r)):
                return r
    return best


def window_dp(can):
    """the same maximum by dynamic programming (any length)."""
    n = len(can)
    h, f = [0] * n, [0] * n
    for p in range(n):
        if can[p]:
            h[p] = 1 + max(h[p - 1] if p and can[p - 1] else 0, f[p - 8] if p >= 8 else 0)
        f[p] = max(f[p - 1] if p else 0, h[p])
    return f[-1] if n else 0


def test_recurrence_equals_exhaustive_search(shim):
    rng = np.random.default_rng(5)
    for _ in range(300):      # pair-aligned masks of 16 windows: device == exhaustive == window DP
        pm = rng.integers(0, 2, size=8)
        can = [int(b) for b in pm for _ in (0, 1)]
        assert device_bound(shim, can, 8) == exhaustive_bound(can) == window_dp(can), can
    for _ in range(300):      # ragged masks: the window DP is exact, the device bound never below it
        can = [int(b) for b in rng.integers(0, 2, size=16)]
        assert window_dp(can) == exhaustive_bound(can)
        assert device_bound(shim, can, 8) >= window_dp(can)
    for npairs in (20, 34, 37, 61, 77):
        for _ in range(400):
            p = rng.choice([0.3, 0.5, 0.8])
            pm = (rng.random(npairs) < p).astype(int)
            can = [int(b) for b in pm for _ in (0, 1)]
            assert device_bound(shim, can, npairs) == window_dp(can)
            ragged = [int(c and rng.random() < 0.8) for c in can]
            assert device_bound(shim, ragged, npairs) >= window_dp(ragged)


def test_the_pattern_the_kernel_asks(shim):
    """150-base read, 68 windows: even pairs ruled out -> 12 votes at most; each pair the filter lets through adds 3."""
    can = [0] * 68
    for P in range(34):
        if P % 2 == 1:
            can[2 * P] = can[2 * P + 1] = 1
    assert device_bound(shim, can, 34) == 12
    can[2 * 10] = can[2 * 10 + 1] = 1    # one false positive: a run of 6 windows
    assert device_bound(shim, can, 34) == 15
    assert device_bound(shim, [1] * 68, 34) == 68
    assert device_bound(shim, [0] * 68, 34) == 0


def _votes_and_presence(oracle, ox, read: bytes):
    """first pass of indexer.rs:275-321 spelled out: per stride-2 window its in-table flag, per diagonal its voters."""
    nwin = (len(read) - 16) // 2 + 1 if len(read) >= 16 else 0
    present = [0] * nwin
    voters = {}
    for w in range(nwin):
        k = oracle.make_kmer(read, 2 * w)
        if k < 0:
            continue
        n, sites = ox.lookup(k)
        if n == 0:
            continue
        present[w] = 1   # unique, 2..5-fold or HIGH: the key is in the table
        if n < 0:
            continue     # HIGH: no votes
        for (c, p) in sites:
            voters.setdefault(oracle.gp_to_i64(c, p - 2 * w), []).append(w)
    return present, voters


def _check_reads(oracle, shim, ox, reads):
    n_checked = 0
    for read in reads:
        present, voters = _votes_and_presence(oracle, ox, read)
        if not present:
            continue
        npairs = (len(present) + 1) // 2
        fit = min(x for x in (8, 20, 34, 37, 61, 77, 10 ** 9) if x >= npairs)
        bound_all = device_bound(shim, present, fit) if fit < 10 ** 9 else window_dp(present)
        for diag, V in voters.items():
            V = sorted(set(V))
            for a, b in zip(V, V[1:]):
                if b - a <= 7:
                    assert all(present[a:b + 1]), (read, diag, a, b)
            assert len(V) <= window_dp(present) <= bound_all, (read, diag)
            n_checked += 1
    return n_checked


def test_statement_holds_for_the_oracles_votes_branch_cases(oracle, shim):
    genes, _ = branch_genes()
    ox = oracle.OracleIndexer(genes)
    reads = [r for _, r in branch_reads(genes)]
    assert _check_reads(oracle, shim, ox, reads) > 300


def test_statement_holds_on_strand_junctions_and_tiny_genes(oracle, shim):
    """reads laid across the point where a gene's reverse strand meets its forward strand in site-code space
    (rc(gene) + gene), across gene ends, over genes of 16..40 bases and over tandem repeats."""
    rng = np.random.default_rng(23)
    genes = [rand_seq(rng, int(n)) for n in (16, 17, 18, 24, 33, 40, 64, 150, 400, 1200)]
    unit = rand_seq(rng, 7)
    genes.append((unit * 60)[:400])                      # tandem repeat: 2..5-fold and HIGH keys, shifted diagonals
    g = bytearray(rand_seq(rng, 600)); g[300] = ord("N"); genes.append(bytes(g))
    ox = oracle.OracleIndexer(genes)
    reads = []
    for g in genes:
        both = rc(g) + g        # the reverse strand followed by the forward strand, as gdu lays them out
        both2 = g + rc(g)
        for src in (both, both2, rc(g)[:-1] + g, g + g, rc(g) + rc(g)):   # (rc(g)[:-1] + g is gdu's own layout)
            for L in (40, 80, 150):
                for _ in range(6):
                    if len(src) <= L:
                        reads.append(src)
                        continue
                    s = int(rng.integers(0, len(src) - L))
                    r = bytearray(src[s:s + L])
                    if rng.random() < 0.5:
                        r[int(rng.integers(0, L))] = b"ACGT"[int(rng.integers(0, 4))]
                    reads.append(bytes(r))
    for _ in range(200):        # chimeras of two genes, any strands
        a, b = genes[int(rng.integers(0, len(genes)))], genes[int(rng.integers(0, len(genes)))]
        a = rc(a) if rng.random() < 0.5 else a
        b = rc(b) if rng.random() < 0.5 else b
        reads.append((a[-75:] + b[:75]))
    assert _check_reads(oracle, shim, ox, reads) > 1000


def test_high_keys_stand_for_the_bound(oracle, shim):
    """A window whose key has six sites or more (HIGH) cannot vote, but it IS in the table: it may sit between two
    voters of one diagonal.  Striking it from the standing windows — as "cannot vote" suggests — splits their run and
    makes the bound too small: here a read with count1 = 20 and count2 = 10 (it passes the gate of indexer.rs:353-360)
    would get 19.  gf_k_seedverify_stream therefore keeps HIGH seeds standing (khigh).  Found by enumeration, r03."""
    rng = np.random.default_rng(314)
    H = rand_seq(rng, 16)                       # the 16-mer that will be HIGH
    dump = b"".join(H + rand_seq(rng, 40) for _ in range(7))   # seven copies elsewhere
    ga = bytearray(rand_seq(rng, 400))
    gb = bytearray(rand_seq(rng, 400))
    # read layout (stride-2 windows): run A = windows 2..22 with the HIGH key at window 16, run B = windows 43..53 with
    # it at window 48; everything else random
    pa, pb = 100, 150
    ga[pa + 2 * (16 - 2):pa + 2 * (16 - 2) + 16] = H          # read base 32 = gene A base pa + 28
    gb[pb + 2 * (48 - 43):pb + 2 * (48 - 43) + 16] = H        # read base 96 = gene B base pb + 10
    genes = [bytes(ga), bytes(gb), dump]
    ox = oracle.OracleIndexer(genes)
    n_h, _ = ox.lookup(oracle.make_kmer(H, 0))
    assert n_h == -2                                           # HIGH
    read = bytearray(rand_seq(rng, 150))
    read[4:4 + (2 * (22 - 2) + 16)] = genes[0][pa:pa + 2 * 20 + 16]      # windows 2..22: bases 4..59
    read[86:86 + (2 * (53 - 43) + 16)] = genes[1][pb:pb + 2 * 10 + 16]   # windows 43..53: bases 86..121
    read = bytes(read)
    present, voters = _votes_and_presence(oracle, ox, read)
    counts = sorted((len(set(v)) for v in voters.values()), reverse=True)
    assert counts[0] == 20 and counts[1] == 10                 # passes the first gate: 2*20 >= 40, 2*10 >= 20
    assert present[16] == 1 and present[48] == 1              # the HIGH windows are in the table
    assert device_bound(shim, present, 34) >= 20               # the bound over what is in the table holds
    struck = list(present)
    struck[16] = struck[48] = 0                                # "cannot vote" mistaken for "absent"
    assert window_dp(struck) == 19                             # ... would have proved a read dead that passes the gate
    assert ox.map_read(read) == []                             # (the mismatch gate ends it later: too much of it is random)
