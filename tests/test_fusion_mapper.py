"""FusionMapper::map_read tail (mapable rule, direction gate, make_match,
calc_distance/calc_ed, edit_distance).  Host logic: the product's C ABI entry points are
compared with the C++ oracle and the independent Python model on the CPU; the end-to-end
form (GPU mapping + host tail + reverse-complement retry) is a GPU test."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from oracle import indexer_model as M
from tests.helpers import rand_seq, rc

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "branch_cases.json")


def test_edit_distance_three_ways(oracle):
    """edit_distance.rs:221-261 prints three distances (0, 1, 90 per SURVEY.md §4); the
    bit-parallel routine, its restatement and the textbook DP must agree everywhere, across
    the block boundaries (64, 128, 640 symbols) and the DP fallback (> 640)."""
    from genefuserust_amd import edit_distance
    rng = np.random.default_rng(0)
    assert edit_distance(b"", b"ACGT") == 4 == oracle.edit_distance(b"", b"ACGT")
    assert edit_distance(b"ACGT", b"") == 4
    assert edit_distance(b"kitten", b"sitting") == 3 == oracle.edit_distance(b"kitten", b"sitting")
    for la, lb in [(1, 1), (5, 70), (63, 64), (64, 64), (65, 64), (127, 129), (128, 200), (150, 150), (75, 75),
                   (300, 310), (640, 640), (641, 30), (700, 650), (900, 1000)]:
        for rep in range(3):
            a = rand_seq(rng, la)
            b = bytearray(a[:lb] if rep else rand_seq(rng, lb))
            if len(b) < lb:
                b += rand_seq(rng, lb - len(b))
            for _ in range(int(rng.integers(0, 6))):
                b[int(rng.integers(0, len(b)))] = ord("N")
            b = bytes(b)
            want = M.levenshtein(a.decode(), b.decode())
            assert oracle.edit_distance(a, b) == want, (la, lb, rep)
            assert edit_distance(a, b) == want, (la, lb, rep)
            assert edit_distance(b, a) == want


def test_edit_distance_reference_vectors(oracle):
    """The three pairs with expected distances 0 / 1 / 90 that the reference's own unit test holds
    (edit_distance.rs:221-261), kept as data in tests/golden/edit_distance_ref_test.json: a pin
    of gf_edit_distance, of the oracle's restatement and of the independent DP model."""
    from genefuserust_amd import edit_distance
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "edit_distance_ref_test.json")))
    assert [c["expect"] for c in fx["cases"]] == [0, 1, 90]
    for c in fx["cases"]:
        a, b = c["a"].encode(), c["b"].encode()
        assert edit_distance(a, b) == c["expect"] == edit_distance(b, a)
        assert oracle.edit_distance(a, b) == c["expect"]
        assert M.levenshtein(c["a"], c["b"]) == c["expect"]


def _c_tail(fusion_seq, rev, read, mapping):
    from genefuserust_amd import _lib
    L = _lib.lib()
    n = len(fusion_seq)
    seqs = [s.encode() for s in fusion_seq]
    arr = (C.c_char_p * max(n, 1))(*seqs)
    lens = (C.c_int64 * max(n, 1))(*[len(s) for s in seqs])
    revb = np.asarray(list(rev) or [0], dtype=np.uint8)
    mm = (_lib.GfSeqMatch * max(len(mapping), 1))()
    for k, m in enumerate(mapping):
        mm[k] = _lib.GfSeqMatch(m[0], m[1], m[3], m[2], 0)
    out = _lib.GfReadMatch()
    st = _lib.check(L.gf_fusion_map_read(arr, lens, n, revb.ctypes.data, read, len(read), mm, len(mapping), C.byref(out)))
    return st, ({k: int(getattr(out, k)) for k, _ in _lib.GfReadMatch._fields_} if st == 2 else None)


def test_tail_on_golden_mappings(oracle):
    """Every golden read with its golden Vec<SeqMatch>: product == oracle == model."""
    g = json.load(open(GOLDEN))
    genes = [None if x is None else x.encode() for x in g["genes"]]
    ox = oracle.OracleIndexer(genes)
    fusion_seq = [ox.fusion_seq(c) for c in range(len(genes))]
    rev = g["reversed"]
    n_match = 0
    for c in g["cases"]:
        mapping = [tuple(m) for m in c["expect"]]
        read = c["read"].encode()
        want = oracle.fusion_map_read(ox, rev, read, mapping)
        assert M.fusion_map_read(fusion_seq, rev, c["read"], mapping) == want, c["label"]
        assert _c_tail(fusion_seq, rev, read, mapping) == want, c["label"]
        n_match += want[0] == 2
    assert n_match >= 20


def test_tail_known_answer_and_quirks(oracle):
    """Planted fusion: read_break 74, left_gp=(0,p), right_gp=(1,q), gap 0,
    both distances 0; the strand and range quirks of calc_ed (-1, -2)."""
    rng = np.random.default_rng(3)
    genes = [rand_seq(rng, 1200), rand_seq(rng, 1100)]
    ox = oracle.OracleIndexer(genes)
    fs = [g.decode() for g in genes]
    p, q = 500, 300
    read = genes[0][p - 74:p + 1] + genes[1][q:q + 75]
    mapping = [(0, 74, 0, p - 74), (75, 149, 1, q - 75)]
    want = (2, {"read_break": 74, "gap": 0, "left_distance": 0, "right_distance": 0, "left_position": p,
                "right_position": q, "left_contig": 0, "right_contig": 1})
    assert oracle.fusion_map_read(ox, [False, False], read, mapping) == want
    assert M.fusion_map_read(fs, [False, False], read.decode(), mapping) == want
    assert _c_tail(fs, [False, False], read, mapping) == want
    # one substitution on each side
    r2 = bytearray(read); r2[10] = ord("A") if r2[10] != ord("A") else ord("C"); r2[100] = ord("N")
    st, rm = _c_tail(fs, [False, False], bytes(r2), mapping)
    assert (st, rm["left_distance"], rm["right_distance"]) == (2, 1, 1)
    assert oracle.fusion_map_read(ox, [False, False], bytes(r2), mapping) == (st, rm)
    # both negative -> not in required direction -> mapable, no match (caller retries the RC)
    neg = [(75, 149, 0, -(p + 75)), (0, 74, 1, -(q + 74))]
    for f in (oracle.fusion_map_read(ox, [False, False], rc(read), neg), _c_tail(fs, [False, False], rc(read), neg),
              M.fusion_map_read(fs, [False, False], rc(read).decode(), neg)):
        assert f == (1, None)
    # fewer than two segments -> unmapable
    assert _c_tail(fs, [False, False], read, mapping[:1]) == (0, None)
    # calc_ed quirks: a segment whose start is exactly 0 counts as "different strands" (-1);
    # a segment running off the gene gives -2
    m0 = [(0, 74, 0, 0), (75, 149, 1, q - 75)]       # left start = 0
    st, rm = _c_tail(fs, [False, True], genes[0][0:75] + genes[1][q:q + 75], m0)
    assert st == 2 and rm["left_distance"] == -1
    assert oracle.fusion_map_read(ox, [False, True], genes[0][0:75] + genes[1][q:q + 75], m0) == (st, rm)
    far = [(0, 74, 0, 1150), (75, 149, 1, q - 75)]   # left end beyond the gene
    st, rm = _c_tail(fs, [False, False], read, far)
    assert st == 2 and rm["left_distance"] == -2
    assert oracle.fusion_map_read(ox, [False, False], read, far) == (st, rm)


@pytest.mark.gpu
def test_fusion_mapper_end_to_end(gpu_device, oracle):
    """GPU mapping + host tail + reverse-complement retry on the golden reads: the
    reference's per-read policy (map, else retry the RC when mapable) reproduced with two
    batched GPU calls."""
    from genefuserust_amd import Fusion, FusionMapper, Gene, Indexer
    g = json.load(open(GOLDEN))
    genes = [None if x is None else x.encode() for x in g["genes"]]
    ix = Indexer.from_gene_slices(genes, g["reversed"])
    ix.make_index()
    fm = FusionMapper(ix)
    ox = oracle.OracleIndexer(genes)
    reads = [c["read"].encode() for c in g["cases"]]
    got = fm.scan_single_end(reads)
    n_match = n_retry = 0
    for read, res in zip(reads, got):
        st, rm = oracle.fusion_map_read(ox, g["reversed"], read, ox.map_read(read))
        reversed_ = False
        if st == 1:  # mapable but no match: retry the reverse complement
            r2 = rc(read)
            st, rm = oracle.fusion_map_read(ox, g["reversed"], r2, ox.map_read(r2))
            reversed_ = True
            n_retry += 1
        if st != 2:
            assert res is None
            continue
        n_match += 1
        assert res is not None and res.m_reversed == reversed_
        assert (res.m_read_break, res.m_gap, res.m_left_distance, res.m_right_distance) == (
            rm["read_break"], rm["gap"], rm["left_distance"], rm["right_distance"])
        assert (res.m_left_gp, res.m_right_gp) == ((rm["left_contig"], rm["left_position"]),
                                                    (rm["right_contig"], rm["right_position"]))
    assert n_match >= 20 and n_retry >= 5
    # one read at a time gives the same as the batch
    for read in reads[:40]:
        single, mapable = fm.map_read(read)
        st, rm = oracle.fusion_map_read(ox, g["reversed"], read, ox.map_read(read))
        assert (single is not None) == (st == 2) and mapable == (st != 0)
    ix.close()


def test_filter_matches_and_sort_order():
    """fusion_mapper.rs:276-384 without remove_alignables: the product's predicates against the
    independent model on seeded matches that reach every reason, and the sort order on ties."""
    from genefuserust_amd import FusionMapper, ReadMatch
    from genefuserust_amd.indexer import GenePos
    rng = np.random.default_rng(4)
    fm = FusionMapper.__new__(FusionMapper)   # the predicates need no index
    ms, want = [], []
    for k in range(600):
        ln = int(rng.integers(40, 160))
        seq = rand_seq(rng, ln)
        if k % 5 == 0:   # low-complexity flank: a homopolymer run on one side
            cut = int(rng.integers(20, ln - 20))
            seq = (b"A" * cut + seq[cut:]) if k % 2 else (seq[:cut] + b"AC" * 3 + b"T" * (ln - cut - 6))
        brk = int(rng.integers(0, ln - 1)) if k % 7 else int(rng.choice([5, ln - 8]))
        lc, rc_ = int(rng.integers(0, 3)), int(rng.integers(0, 3))
        lp = int(rng.integers(-3000, 3000))
        rp = lp + int(rng.integers(-80, 80)) if k % 3 == 0 else int(rng.integers(-3000, 3000))
        ld, rd = int(rng.integers(-2, 5)), int(rng.integers(-2, 5))
        m = ReadMatch(seq, brk, GenePos(lc, lp), GenePos(rc_, rp), 0, ld, rd, False, b"@r%d" % int(rng.integers(0, 50)))
        ms.append(m)
        want.append(M.match_filter(seq.decode(), brk, (lc, lp), (rc_, rp), ld, rd, 50))
    kept, removed = fm.filter_matches(ms, 50)
    assert [m for m, w in zip(ms, want) if w == 0] == kept
    assert removed == {"complexity": want.count(1), "distance": want.count(2), "indels": want.count(3)}
    assert min(removed.values()) > 10 and len(kept) > 10
    got = FusionMapper.sort_matches(ms)
    exp = M.match_sort([(m.m_read_break, len(m.m_read), m.m_name, i) for i, m in enumerate(ms)])
    assert [(g.m_read_break, len(g.m_read), g.m_name) for g in got] == [(e[0], e[1], e[2]) for e in exp]
    # known answers: break first, then the shorter read, then the larger name
    a = ReadMatch(b"A" * 50, 30, GenePos(0, 1), GenePos(1, 1), 0, 0, 0, False, b"x")
    b = ReadMatch(b"A" * 40, 30, GenePos(0, 1), GenePos(1, 1), 0, 0, 0, False, b"a")
    c = ReadMatch(b"A" * 40, 30, GenePos(0, 1), GenePos(1, 1), 0, 0, 0, False, b"b")
    d = ReadMatch(b"A" * 90, 31, GenePos(0, 1), GenePos(1, 1), 0, 0, 0, False, b"a")
    assert [m.m_name + bytes([len(m.m_read)]) for m in FusionMapper.sort_matches([a, b, c, d])] == \
        [b"a" + bytes([90]), b"b" + bytes([40]), b"a" + bytes([40]), b"x" + bytes([50])]
