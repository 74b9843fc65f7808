"""the reference's Matcher / remove_alignables as they are — the vectorised
restatement (genefuserust_amd/matcher.py) against the literal loop-by-loop model
(oracle/indexer_model.py).  Host only.  Parity unpinned: the reference has no test for it."""
import numpy as np
import pytest

from genefuserust_amd import GenePos, ReadMatch
from genefuserust_amd.matcher import Matcher, MatcherPanic, remove_alignables
from oracle import indexer_model as M
from tests.helpers import rand_seq


def genome(rng, n_contigs, length, n_gaps, polya):
    out = {}
    for c in range(n_contigs):
        s = bytearray(rand_seq(rng, length))
        for _ in range(n_gaps):  # N gaps: the index restarts after each
            p = int(rng.integers(0, length - 40))
            s[p:p + int(rng.integers(1, 30))] = b"N" * 29
            s = s[:length]
        for _ in range(polya):  # poly-A runs: every position after 15 of them is filed
            p = int(rng.integers(0, length - 60))
            ln = int(rng.integers(10, 40))
            s[p:p + ln] = b"A" * ln
        if c % 2:
            s = s.lower()  # the Matcher upper-cases the contigs itself
        out["chr%d" % (n_contigs - c)] = bytes(s)  # (names out of order: the reference's map is sorted)
    return out


def outcome_model(index, seq):
    try:
        return M.matcher_model_do_match(index, seq)
    except M.ModelPanic:
        return "panic"


def outcome(m, seq):
    try:
        return m.do_match(seq)
    except MatcherPanic:
        return "panic"


@pytest.mark.parametrize("seed,n_contigs,length,gaps,polya", [(1, 2, 600, 0, 0), (2, 3, 2000, 3, 2), (3, 2, 4000, 30, 30),
                                                                (4, 1, 300, 1, 1), (5, 4, 3000, 80, 0)])
def test_matcher_against_the_literal_model(seed, n_contigs, length, gaps, polya):
    rng = np.random.default_rng(seed)
    contigs = genome(rng, n_contigs, length, gaps, polya)
    reads = [rand_seq(rng, int(rng.integers(15, 160))) for _ in range(12)]
    reads += [b"ACGTNNACGT" * 6, b"A" * 40, b"acgtacgtacgtacgtacgt", b"CCCCCCCCCCCCCCCCCCCCGGGGGG"]
    if seed == 4:
        reads = [b"C" * 30, b"G" * 20]  # no window starts with A or T: those keys never enter the index
    m = Matcher(contigs, reads)
    bloom, index = M.matcher_model_build({k: v.decode() for k, v in contigs.items()}, [r.decode() for r in reads])
    assert set(bloom) <= {0} and bloom.get(0, 0) == m.bloom_bits
    assert {k: v for k, v in m.m_kmer_positions.items()} == index
    assert set(index) <= {0, 1, 2, 3}
    assert m.m_contig_names == sorted(contigs)
    for r in reads:
        assert outcome(m, r) == outcome_model(index, r.decode()), r
    # never a match: None, or the reference's panic
    assert {outcome(m, r) for r in reads} <= {None, "panic"}


def test_remove_alignables_is_a_no_op_on_a_genome_with_enough_sites():
    """Every key with more than 50 sites (what N gaps and poly-A runs do to a human genome):
    nothing votes, nothing is removed."""
    rng = np.random.default_rng(9)
    contigs = genome(rng, 3, 6000, 120, 40)
    reads = [rand_seq(rng, 150) for _ in range(20)]
    ms = [ReadMatch(r, 70, GenePos(0, 100), GenePos(1, 200), 0, 0, 0) for r in reads]
    m = Matcher(contigs, reads)
    assert set(m.m_kmer_positions) == {0, 1, 2, 3} and all(len(v) > 50 for v in m.m_kmer_positions.values())
    kept, removed = remove_alignables(ms, contigs)
    assert kept == ms and removed == 0
    assert remove_alignables(ms, None) == (ms, 0)  # no reference loaded: fusion_mapper.rs:489-491
    assert remove_alignables([], contigs) == ([], 0)


def test_small_reference_panics_like_the_reference():
    """Two contigs that start with A and C: keys 0 and 2 have one site each, they vote, and the
    first window of the read that starts with T or G hits the unwrap."""
    contigs = {"a": b"ACGTTGCAAGGCTTAACCGGTTAACC", "b": b"CGTTGCAAGGCTTAACCGGTTAACCA"}
    read = b"ACGTTGCAAGGCTTAACCGGTTAACCGGATCGATCG"
    with pytest.raises(MatcherPanic):
        Matcher(contigs, [read]).do_match(read)
    with pytest.raises(M.ModelPanic):
        _, index = M.matcher_model_build({k: v.decode() for k, v in contigs.items()}, [read.decode()])
        M.matcher_model_do_match(index, read.decode())
    with pytest.raises(MatcherPanic):
        Matcher(contigs, [b"ACGT"])  # a candidate shorter than 15 bases: the range underflows
    # a read whose windows all start with A or C finds both keys in the index: no panic, no match
    assert Matcher(contigs, [b"ACACACACACACACACACACAAAA"]).do_match(b"ACACACACACACACACACACAAAA") is None


@pytest.mark.parametrize("seed,n_contigs,length,gaps,polya", [(1, 2, 600, 0, 0), (2, 3, 2000, 3, 2), (3, 2, 4000, 30, 30),
                                                                (5, 4, 3000, 80, 0), (7, 2, 40, 0, 0)])
def test_cpp_matcher_mirror(tmp_path, seed, n_contigs, length, gaps, polya):
    """include/gf_matcher.hpp (the C++ form of the same as-is behaviour) builds the same index and gives
    the same outcome per read as the Python mirror, which the literal model checks above."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "test_matcher")
    subprocess.run(["g++", "-std=c++17", "-O1", "-I" + os.path.join(root, "include"),
                    os.path.join(root, "tests", "cpp", "test_matcher.cpp"), "-o", exe], check=True)
    rng = np.random.default_rng(seed)
    if length < 100:   # a small reference: keys with few sites vote, reads panic
        contigs = {"a": b"ACGTTGCAAGGCTTAACCGGTTAACC", "b": b"CGTTGCAAGGCTTAACCGGTTAACCA"}
        reads = [b"ACGTTGCAAGGCTTAACCGGTTAACCGGATCGATCG", b"ACACACACACACACACACACAAAA", rand_seq(rng, 60)]
    else:
        contigs = genome(rng, n_contigs, length, gaps, polya)
        reads = [rand_seq(rng, int(rng.integers(15, 160))) for _ in range(12)]
        reads += [b"ACGTNNACGT" * 6, b"A" * 40, b"acgtacgtacgtacgtacgt", b"CCCCCCCCCCCCCCCCCCCCGGGGGG"]
    case = tmp_path / "case.txt"
    with open(case, "w") as f:
        f.write("%d %d\n" % (len(contigs), len(reads)))
        for k, v in contigs.items():
            f.write("%s %s\n" % (k, v.decode()))
        for r in reads:
            f.write(r.decode() + "\n")
    out = subprocess.run([exe, str(case)], capture_output=True, text=True, check=True).stdout.split("\n")
    m = Matcher(contigs, reads)
    assert out[0] == "bloom %d" % m.bloom_bits
    assert out[1].split()[1:] == m.m_contig_names
    keys = sorted(m.m_kmer_positions)
    for j, k in enumerate(keys):
        want = "key %d %d" % (k, len(m.m_kmer_positions[k])) + "".join(" %d:%d" % s for s in m.m_kmer_positions[k])
        assert out[2 + j] == want, k
    got = out[2 + len(keys):2 + len(keys) + len(reads)]
    assert got == [("panic" if outcome(m, r) == "panic" else "none") for r in reads]
    try:
        kept, _ = remove_alignables([ReadMatch(r, 70, GenePos(0, 100), GenePos(1, 200), 0, 0, 0) for r in reads], contigs)
        want_kept = "kept %d" % len(kept)
    except MatcherPanic:
        want_kept = "kept panic"
    assert out[2 + len(keys) + len(reads)] == want_kept
