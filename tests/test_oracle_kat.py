"""Known answers derived by hand from the reference source,
checked on BOTH restatements (oracle/indexer_oracle.cc and oracle/indexer_model.py).
The reference's own tests hold no expected value for this path (parity unpinned);
these are the pins the build created."""
import numpy as np
import pytest

from oracle import indexer_model as M
from tests.helpers import rand_seq, rc


def test_kmer_values(oracle):
    # first 16 bases of the 42-mer at indexer.rs:1073
    s = b"CATCACACACCTTGACTGGTCCCCAGACAACAAGTATATAAT"
    k0 = oracle.make_kmer(s, 0, -1, 2)
    assert k0 == 2250385778 == 0x86222972
    assert M.kmer_at(s.decode(), 0) == k0
    # next stride-2 window via the rolling form equals the from-scratch value
    k2 = oracle.make_kmer(s, 2, k0, 2)
    assert k2 == 0x62229727 == oracle.make_kmer(s, 2, -1, 2) == M.kmer_at(s.decode(), 2)
    assert k2 == ((k0 & 0x0FFFFFFF) << 4) | (1 << 2) | 3  # ...T G
    assert oracle.make_kmer(b"A" * 16, 0) == 0
    assert oracle.make_kmer(b"G" * 16, 0) == 0xFFFFFFFF
    assert oracle.make_kmer(b"ACGTNACGTACGTACGT", 0) == -1
    assert oracle.make_kmer(b"acgtacgtacgtacgt", 0) == -1
    # rolling over every step agrees with scratch
    rng = np.random.default_rng(1)
    t = rand_seq(rng, 80)
    for step in (1, 2):
        last = -1
        for i in range(0, 80 - 16 + 1, step):
            last = oracle.make_kmer(t, i, last, step)
            assert last == oracle.make_kmer(t, i, -1, step) == M.kmer_at(t.decode(), i)


def test_key64_roundtrip(oracle):
    assert oracle.gp_to_i64(0, 5) == 5 == M.key64(0, 5)
    assert oracle.gp_to_i64(1, -20) == 0x1FFFFFFEC == 8589934572 == M.key64(1, -20)
    assert oracle.gp_to_i64(3, 0) == 12884901888 == M.key64(3, 0)
    # the ten pairs of indexer.rs:982-983 (contig -1 sign-extends over the whole word)
    contigs = [0, 1, 3, 220, -1, 0, 23, 4440, 110, 10]
    positions = [0, 111, 222, -333, 444, 555555, 6, -7777777, 8888, -9999]
    for c, p in zip(contigs, positions):
        v = oracle.gp_to_i64(c, p)
        assert oracle.i64_to_gp(v) == (c, p)
        if c >= 0:
            assert M.unkey64(M.key64(c, p)) == (c, p)


def test_reverse_complement(oracle):
    # sequence.rs:67-70
    assert oracle.reverse_complement(b"ATGCGGGTT") == b"AACCCGCAT"
    assert oracle.reverse_complement(b"CGAANTAG") == b"CTANTTCG"
    assert M.revcomp("ATGCGGGTT") == "AACCCGCAT" and M.revcomp("CGAANTAG") == "CTANTTCG"
    assert oracle.reverse_complement(b"acgtx") == b"NACGT" == rc(b"acgtx")


def _two_unique_genes(seed=3):
    rng = np.random.default_rng(seed)
    return [rand_seq(rng, 1200), rand_seq(rng, 1100)]


def test_probe_counts():
    L = 150
    assert len(range(0, L - 16 + 1, 2)) == 68 and L - 16 + 1 == 135


@pytest.mark.parametrize("p,q", [(500, 300), (74 + 1, 75 + 1), (1100, 900)])
def test_planted_fusion(oracle, p, q):
    genes = _two_unique_genes()
    # the hand derivation assumes the two halves do not extend by chance: the base
    # after the left part differs from the right part's first base and vice versa
    while genes[0][p + 1] == genes[1][q] or genes[1][q - 1] == genes[0][p]:
        p, q = p + 1, q + 1
    read = genes[0][p - 74:p + 1] + genes[1][q:q + 75]
    assert len(read) == 150
    expect = [(0, 74, 0, p - 74), (75, 149, 1, q - 75)]
    ox = oracle.OracleIndexer(genes)
    assert ox.map_read(read) == expect
    mx = M.IndexModel([g.decode() for g in genes])
    assert mx.map_read(read.decode()) == expect
    # the reverse-complemented read: TOP is the right half on contig 0
    expect_rc = [(75, 149, 0, -(p + 75)), (0, 74, 1, -(q + 74))]
    assert ox.map_read(rc(read)) == expect_rc
    assert mx.map_read(rc(read).decode()) == expect_rc
    # downstream make_match arithmetic (fusion_mapper.rs:160-189)
    left, right = expect
    read_break = (left[1] + right[0]) // 2
    assert read_break == 74 and left[3] + read_break == p and right[3] + read_break + 1 == q
    assert M.in_required_direction(expect, [False, False]) is True
    assert M.in_required_direction(expect_rc, [False, False]) is False
    assert oracle.in_required_direction(expect, [False, False]) is True
    assert oracle.in_required_direction(expect_rc, [False, False]) is False


def test_left_diagonal_zero_is_invisible(oracle):
    genes = _two_unique_genes()
    read = genes[0][0:75] + genes[1][300:375]  # left diagonal = (0, 0) = key 0
    assert oracle.OracleIndexer(genes).map_read(read) == []
    assert M.IndexModel([g.decode() for g in genes]).map_read(read.decode()) == []


def test_short_and_invalid_reads(oracle):
    genes = _two_unique_genes()
    ox = oracle.OracleIndexer(genes)
    mx = M.IndexModel([g.decode() for g in genes])
    for ln in (0, 1, 15, 16, 53):
        r = genes[0][100:100 + ln]
        assert ox.map_read(r) == [] and mx.map_read(r.decode()) == []
    fusion = genes[0][426:501] + genes[1][300:375]
    for r in (b"N" * 150, fusion.lower()):
        assert ox.map_read(r) == [] and mx.map_read(r.decode()) == []


def test_dupe_classes(oracle):
    rng = np.random.default_rng(5)
    elem = rand_seq(rng, 16)
    for copies, expect_n in ((1, 1), (2, 2), (5, 5), (6, -2), (9, -2)):
        g = bytearray(rand_seq(rng, 400))
        for k in range(copies):
            g[20 + 40 * k:36 + 40 * k] = elem
        ox = oracle.OracleIndexer([bytes(g)])
        n, sites = ox.lookup(oracle.make_kmer(elem, 0))
        assert n == expect_n
        if n > 0:
            assert sites == [(0, 20 + 40 * k) for k in range(copies)]
        mx = M.IndexModel([bytes(g).decode()])
        v = mx.table[M.kmer_at(elem.decode(), 0)]
        assert (v is M.HIGH) == (expect_n == -2)


def test_last_window_not_indexed(oracle):
    rng = np.random.default_rng(9)
    g = rand_seq(rng, 100)
    ox = oracle.OracleIndexer([g])
    assert ox.lookup(oracle.make_kmer(g, 83))[0] == 1   # i = len-17 is the last forward window
    assert ox.lookup(oracle.make_kmer(g, 84))[0] == 0   # i = len-16 is never indexed (indexer.rs:188)
    r = rc(g)
    n, sites = ox.lookup(oracle.make_kmer(r, 0))
    assert (n, sites) == (1, [(0, 1 - 100)])
    assert ox.lookup(oracle.make_kmer(r, 84))[0] == 0
    g17 = rand_seq(rng, 17)
    ox17 = oracle.OracleIndexer([g17])
    assert ox17.stats()["n_keys"] == 2
    assert oracle.OracleIndexer([rand_seq(rng, 16)]).stats()["n_keys"] == 0


def test_segment_mask_quirks(oracle):
    # a run that begins at the last base is never seen (indexer.rs:635-640)
    m = [3] * 30 + [0] * 119 + [2]
    assert oracle.segment_mask(m, (0, 5), (1, 7)) == [(0, 29, 0, 5)] == M.segment_mask(m, (0, 5), (1, 7))
    # gaps < 10 are bridged, a 10-gap is not; SECOND stops at a TOP cell
    m = [3] * 25 + [0] * 9 + [3] * 5 + [0] * 10 + [3] * 30 + [2] * 11 + [3] + [2] * 30
    m += [0] * (150 - len(m))
    got = oracle.segment_mask(m, (0, 1), (0, 2))
    assert got == M.segment_mask(m, (0, 1), (0, 2))
    assert got[0][:2] == (0, 38)          # first of the two equal-length... longest TOP run
    # threshold: end - start must exceed 20
    m = [3] * 21 + [0] * 20 + [2] * 22 + [0] * 87
    assert oracle.segment_mask(m, (0, 1), (0, 2)) == [(41, 62, 0, 2)] == M.segment_mask(m, (0, 1), (0, 2))


def test_config1_empty_index(oracle):
    # BASELINE config 1: tinyref.fa has none of the fusion genes' chromosomes ->
    # every gene unresolved -> empty index -> every read maps to []
    ox = oracle.OracleIndexer([None, None, None, None])
    assert ox.stats()["n_keys"] == 0
    r1 = (b"CATCACACACCTTGACTGGTCCCCAGACAACAAGTATATAATGTCTAACTCGGGAGACTATGAAATATTGTACTGTAAGTATGAATGATT"
          b"TTATATATATATATATATGCTATGATTATATTTATATATATAATAATTATTTTCCATATAT")
    assert ox.map_read(r1) == []
    assert [ox.fusion_seq(c) for c in range(4)] == ["", "", "", ""]
