"""Shared builders for the parity tests: a small gene set and a read list that
together reach every branch of Indexer::make_index / map_read / segment_mask
(SURVEY.md Appendix A/B/C)."""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = bytes.maketrans(b"ACGTacgt", b"TGCATGCA")


def rand_seq(rng, n: int) -> bytes:
    return ACGT[rng.integers(0, 4, size=n)].tobytes()


def rc(s: bytes) -> bytes:
    """reverse complement as sequence.rs:22-60 does it (non-ACGT -> N)."""
    t = bytes((c if c in b"ACGTacgt" else ord("N")) for c in s)
    return t.translate(_COMP)[::-1].replace(b"\x00", b"N")


def branch_genes(seed: int = 7) -> Tuple[List[Optional[bytes]], List[bool]]:
    """6 genes: three real ones with planted 2x / 5x / 6x repeats, an N, a
    lower-case stretch; one missing (None), one too short to index, one of
    exactly 17 bases (one forward and one reverse window)."""
    rng = np.random.default_rng(seed)
    g0 = bytearray(rand_seq(rng, 3000))
    g1 = bytearray(rand_seq(rng, 2500))
    g2 = bytearray(rand_seq(rng, 2000))
    e5 = rand_seq(rng, 60)   # present 5 times -> every k-mer kept with 5 sites
    e6 = rand_seq(rng, 60)   # present 6 times -> HIGH
    e2 = rand_seq(rng, 80)   # present twice
    for g, p in ((g0, 500), (g0, 1500), (g1, 300), (g1, 1200), (g2, 400)):
        g[p:p + 60] = e5
    for g, p in ((g0, 700), (g0, 1700), (g0, 2700), (g1, 100), (g1, 1800), (g2, 900)):
        g[p:p + 60] = e6
    for g, p in ((g0, 2200), (g2, 1500)):
        g[p:p + 80] = e2
    g1[2000] = ord("N")
    g2[1000:1100] = bytes(g2[1000:1100]).lower()
    short = rand_seq(rng, 10)
    g17 = rand_seq(rng, 17)
    genes: List[Optional[bytes]] = [bytes(g0), bytes(g1), bytes(g2), None, short, g17]
    reversed_flags = [False, True, False, False, True, False]
    return genes, reversed_flags


def branch_reads(genes: List[Optional[bytes]], seed: int = 11) -> List[Tuple[str, bytes]]:
    """(label, read) pairs; see SURVEY.md Appendix B for the hand-derived ones."""
    rng = np.random.default_rng(seed)
    g0, g1, g2 = genes[0], genes[1], genes[2]
    g2u = g2.upper()
    R: List[Tuple[str, bytes]] = []
    p, q = 1000, 700
    fusion = g0[p - 74:p + 1] + g1[q:q + 75]
    R.append(("planted_fusion", fusion))
    R.append(("planted_fusion_rc", rc(fusion)))
    R.append(("left_diag_zero_contig0", g0[0:75] + g1[q:q + 75]))
    R.append(("tie_right_half_smaller_key", g1[q:q + 75] + g0[p:p + 75]))
    for ln in (0, 1, 15, 16, 17, 53, 54, 55):
        R.append(("len_%d" % ln, g0[200:200 + ln]))
    R.append(("all_N", b"N" * 150))
    R.append(("lower_case", fusion.lower()))
    R.append(("mixed_case_tail", fusion[:100] + fusion[100:].lower()))
    R.append(("high_dupe_left", g0[660:760][-75:] + g1[q:q + 75]))
    R.append(("five_fold_left", g0[470:545] + g1[q:q + 75]))
    R.append(("five_fold_both", g0[480:555] + g1[1190:1265]))
    R.append(("two_fold_left", g0[2190:2265] + g1[q:q + 75]))
    R.append(("deletion_20_left_of_break", g0[p - 75:p - 20] + g0[p - 19:p + 1] + g1[q:q + 75]))
    R.append(("insertion_20_left_of_break", g0[p - 73:p - 20] + b"A" + g0[p - 20:p + 1] + g1[q:q + 75]))
    R.append(("len_148", g0[p - 73:p + 1] + g1[q:q + 74]))
    R.append(("len_151", g0[p - 75:p + 1] + g1[q:q + 75]))
    R.append(("len_270_two_parts", g0[p - 119:p + 1] + g1[q:q + 150]))
    R.append(("len_270_three_parts", g0[p - 89:p + 1] + g1[q:q + 90] + g2u[1200:1290]))
    withn = bytearray(fusion)
    withn[40] = ord("N")
    R.append(("N_at_40", bytes(withn)))
    withn[110] = ord("N")
    R.append(("N_at_40_and_110", bytes(withn)))
    R.append(("single_gene_fwd", g0[1200:1350]))
    R.append(("single_gene_rc", rc(g1[1500:1650])))
    R.append(("too_many_mismatches", g0[p - 66:p + 1] + g1[q:q + 68] + rand_seq(rng, 15)))
    R.append(("ten_mismatches_ok", g0[p - 69:p + 1] + g1[q:q + 70] + rand_seq(rng, 10)))
    for right in (25, 32, 34, 36, 40):
        R.append(("right_part_%d" % right, g0[p - (150 - right) + 1:p + 1] + g1[q:q + right]))
    R.append(("negative_forward_diagonal", rand_seq(rng, 8) + g0[0:71] + g1[500:571]))
    # contig-boundary adjacency: key64 (0,-1) = 0xFFFFFFFF and (1,0) = 0x100000000 differ by 1
    R.append(("adjacent_contig_keys", g1[0:75] + g0[74:149]))
    R.append(("lower_case_gene_region", g2u[960:1035] + g0[p:p + 75]))
    R.append(("gene_with_N", g1[1960:2035] + g0[p:p + 75]))
    R.append(("g17_only", genes[5]))
    R.append(("rc_junction_mixed_strand", g0[p - 74:p + 1] + rc(g1[q:q + 75])))
    R.append(("rc_both", rc(g0[p - 74:p + 1]) + rc(g1[q:q + 75])))
    R.append(("same_gene_two_loci", g0[300:375] + g0[2400:2475]))
    R.append(("same_gene_adjacent_diagonals", g0[300:375] + g0[376:451]))
    R.append(("run_at_last_base", g0[p - 74:p + 1] + g1[q:q + 60] + rand_seq(rng, 14) + g1[q + 74:q + 75]))
    for k in range(40):
        R.append(("background_%d" % k, rand_seq(rng, 150)))
    # random junctions: random genes, strands, breaks, lengths
    real = [g0, g1, g2u]
    for k in range(160):
        a, b = rng.integers(0, 3, size=2)
        L = int(rng.choice([100, 148, 150, 151, 200, 250]))
        brk = int(rng.integers(20, L - 20))
        pa = int(rng.integers(brk, len(real[a]) - 1))
        pb = int(rng.integers(0, len(real[b]) - (L - brk)))
        left = real[a][pa - brk + 1:pa + 1]
        right = real[b][pb:pb + L - brk]
        if rng.random() < 0.3:
            left = rc(left)
        if rng.random() < 0.3:
            right = rc(right)
        read = bytearray(left + right)
        for _ in range(int(rng.integers(0, 4))):  # a few substitutions
            read[int(rng.integers(0, L))] = ACGT[rng.integers(0, 4)]
        read = bytes(read)
        if rng.random() < 0.5:
            read = rc(read)
        R.append(("random_junction_%d" % k, read))
    return R


def matches_to_tuples(counts: np.ndarray, matches: np.ndarray):
    out = []
    for r in range(counts.size):
        out.append([(int(matches[r, k]["seq_start"]), int(matches[r, k]["seq_end"]),
                     int(matches[r, k]["contig"]), int(matches[r, k]["position"]))
                    for k in range(int(counts[r]))])
    return out
