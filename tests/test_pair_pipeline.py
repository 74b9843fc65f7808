"""gf_scan_pairs_device — the pair policy of PairEndScanner::scan_pair_end (pescanner.rs:427-518)
for a pack resident in HBM, against the oracle-driven restatement of the policy, read by read."""
import json
import os

import numpy as np
import pytest

from tests.helpers import rand_seq, rc

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "branch_cases.json")


def _make_pairs(rng, genes, n, read_len=150, qual=b"E"):
    """Pairs cut from planted fusions (two thirds) or random sequence: fragments of 150..300 bases,
    either strand, some reads with an N or a low quality."""
    g0, g1, g2 = genes[0], genes[1], genes[2]
    pairs = []
    for k in range(n):
        kind = k % 6
        if kind in (0, 1, 2):
            a, b = (g0, g1) if kind != 2 else (g1, g2)
            p, q = int(rng.integers(300, len(a) - 300)), int(rng.integers(300, len(b) - 400))
            frag = a[p - 150:p] + b[q:q + 150]
        elif kind == 3:   # one gene only
            p = int(rng.integers(0, len(g0) - 300))
            frag = g0[p:p + 300]
        else:
            frag = rand_seq(rng, 300)
        lo = int(rng.integers(0, 60))
        flen = int(rng.integers(150, 300 - lo))
        f = frag[lo:lo + flen]
        if k % 2:
            f = rc(f)
        rl = min(read_len, len(f))
        s1, s2 = bytearray(f[:rl]), bytearray(rc(f)[:rl])
        if k % 11 == 0:
            s1[int(rng.integers(0, rl))] = ord("N")
        q1, q2 = bytearray(qual * rl), bytearray(qual * rl)
        if k % 7 == 0:
            q2[int(rng.integers(0, rl))] = ord("#")
        pairs.append((bytes(s1), bytes(q1), bytes(s2), bytes(q2)))
    return pairs


def _reference_policy(oracle, ox, rev, pairs):
    """scan_pair_end restated with the oracle: per pair the list of (source, found_on_rc,
    m_reversed, read, quality, ReadMatch fields) in push order."""
    def ref_map(seq):
        return oracle.fusion_map_read(ox, rev, seq, ox.map_read(seq))

    def one(seq, qual, source):
        st, rm = ref_map(seq)
        if st == 2:
            return [(source, False, False, seq, qual, rm)]
        if st == 1:
            st, rm = ref_map(rc(seq))
            if st == 2:
                return [(source, True, source != 0, rc(seq), qual[::-1], rm)]
        return []

    out, n_merged = [], 0
    for s1, q1, s2, q2 in pairs:
        m = oracle.fast_merge(s1, q1, s2, q2)
        if m is not None:
            n_merged += 1
            out.append(one(m[0], m[1], 0))
        else:
            out.append(one(s1, q1, 1) + one(s2, q2, 2))
    return out, n_merged


@pytest.mark.gpu
@pytest.mark.parametrize("reversed_flags", ["golden", "all_false", "alternating"])
def test_scan_pairs_device_matches_the_reference_policy(gpu_device, oracle, reversed_flags):
    import torch
    from genefuserust_amd import FusionMapper, Indexer
    from genefuserust_amd.read_pair import finish_pair_hits, pack_reads, scan_pairs_device
    g = json.load(open(GOLDEN))
    genes = [None if x is None else x.encode() for x in g["genes"]]
    rev = {"golden": g["reversed"], "all_false": [False] * len(genes),
           "alternating": [bool(i % 2) for i in range(len(genes))]}[reversed_flags]
    ix = Indexer.from_gene_slices(genes, rev)
    ix.make_index()
    fm = FusionMapper(ix)
    ox = oracle.OracleIndexer(genes)
    rng = np.random.default_rng(5)
    pairs = _make_pairs(rng, genes, 1500)
    want, n_merged = _reference_policy(oracle, ox, rev, pairs)
    lb, lo = pack_reads([p[0] for p in pairs]); lq, _ = pack_reads([p[1] for p in pairs])
    rb, ro = pack_reads([p[2] for p in pairs]); rq, _ = pack_reads([p[3] for p in pairs])
    t = [torch.from_numpy(a).cuda() for a in (lb, lq, lo, rb, rq, ro)]
    res = scan_pairs_device(ix, *t, 150, pair_id_base=1000)
    rec, hb, hq, tot = res.download()
    assert tot["overflow"] == 0 and tot["merged_pairs"] == n_merged
    flat = [(p, w) for p, ws in enumerate(want) for w in ws]
    assert tot["hits"] == len(flat) == rec.shape[0]
    assert tot["retried_reads"] >= sum(1 for _, w in flat if w[1])
    n_rc = 0
    for h, (p, (source, on_rc, m_rev, seq, qual, rm)) in zip(rec, flat):
        assert int(h["pair_id"]) == 1000 + p and int(h["source"]) == source
        assert bool(h["flags"] & 1) == on_rc and bool(h["flags"] & 2) == m_rev
        o, ln = int(h["seq_offset"]), int(h["read_len"])
        assert hb[o:o + ln] == seq and hq[o:o + ln] == qual
        n_rc += on_rc
    # the host tail on the records reproduces the reference's ReadMatch fields
    done = finish_pair_hits(fm, rec, hb, hq)
    assert len(done) == len(flat)
    for (pid, m), (p, (source, on_rc, m_rev, seq, qual, rm)) in zip(done, flat):
        assert pid == 1000 + p and m.m_reversed == m_rev and m.m_quality == qual
        assert (m.m_read_break, m.m_gap, m.m_left_distance, m.m_right_distance) == (
            rm["read_break"], rm["gap"], rm["left_distance"], rm["right_distance"])
        assert (m.m_left_gp, m.m_right_gp) == ((rm["left_contig"], rm["left_position"]),
                                                (rm["right_contig"], rm["right_position"]))
    assert len(flat) >= 300 and n_rc >= 50 and n_merged >= 300
    assert {w[0] for _, w in flat} == {0, 1, 2}
    # capacities: too few retry slots / output records are reported, and asking for room repairs it
    small = scan_pairs_device(ix, *t, 150, retry_cap=8).download()[3]
    assert small["overflow"] & 1 and small["retried_reads"] == tot["retried_reads"]
    few = scan_pairs_device(ix, *t, 150, hits_cap=5, bytes_cap=5 * 300).download()
    assert few[3]["overflow"] & 2 and few[3]["hits"] == tot["hits"] and few[0].shape[0] == 5
    assert [int(x) + 1000 for x in few[0]["pair_id"]] == [int(x) for x in rec["pair_id"][:5]]   # (pair_id_base 0 there)
    # an empty pack
    e = [torch.empty(0, dtype=torch.uint8, device="cuda")] * 2 + [torch.zeros(1, dtype=torch.int64, device="cuda")]
    assert scan_pairs_device(ix, *(e + e), 150).download()[3]["hits"] == 0
    ix.close()


@pytest.mark.gpu
def test_device_policy_equals_the_stepwise_policy(gpu_device):
    """The one-call pipeline and the first form (host between the steps) return the same lists —
    ragged read lengths (so that both 150- and 250-base kernels and long merged reads are used)."""
    from genefuserust_amd import FusionMapper, Indexer
    from genefuserust_amd.read_pair import SequenceReadPair, scan_pair_end, scan_pair_end_stepwise
    g = json.load(open(GOLDEN))
    genes = [None if x is None else x.encode() for x in g["genes"]]
    ix = Indexer.from_gene_slices(genes, g["reversed"])
    ix.make_index()
    fm = FusionMapper(ix)
    rng = np.random.default_rng(8)
    raw = _make_pairs(rng, genes, 900)
    pairs = []
    for k, (s1, q1, s2, q2) in enumerate(raw):
        cut1, cut2 = (len(s1) - k % 40, len(s2) - (k * 7) % 50) if k % 3 == 0 else (len(s1), len(s2))
        pairs.append(SequenceReadPair((s1[:cut1], q1[:cut1]), (s2[:cut2], q2[:cut2])))
    a = scan_pair_end(fm, pairs)
    b = scan_pair_end_stepwise(fm, pairs)
    assert sum(len(x) for x in a) >= 150
    for k, (x, y) in enumerate(zip(a, b)):
        assert [(m.m_source, m.m_reversed, m.m_read, m.m_quality, m.m_read_break, m.m_gap, m.m_left_gp, m.m_right_gp,
                 m.m_left_distance, m.m_right_distance, m.m_merge_diff) for m in x] == \
               [(m.m_source, m.m_reversed, m.m_read, m.m_quality, m.m_read_break, m.m_gap, m.m_left_gp, m.m_right_gp,
                 m.m_left_distance, m.m_right_distance, m.m_merge_diff) for m in y], k
    ix.close()


@pytest.mark.gpu
def test_scan_pairs_full_size_properties(gpu_device, oracle):
    """4 M synthetic pairs (fragments N(300,30) cut from the druggable-shaped genes, junction
    fragments planted), BASELINE-size properties of the one-call pipeline: determinism, every
    record re-derived by the oracle from its own read, the merged count equal to a direct merge
    pass, records in push order."""
    import torch
    from genefuserust_amd import Indexer, synth
    from genefuserust_amd.read_pair import fast_merge_device, scan_pairs_device
    n, L = 4_000_000, 150
    genes = synth.make_geneset("IDX-D", scale=0.25)
    ix = Indexer.from_gene_slices(genes.seqs, genes.reversed_flags)
    ix.make_index()
    pr = synth.make_pairs(genes, n, read_len=L, seed=77, device="cuda")
    a = scan_pairs_device(ix, pr.l_bases, pr.l_quals, pr.offsets, pr.r_bases, pr.r_quals, pr.offsets, L)
    rec, hb, hq, tot = a.download()
    b = scan_pairs_device(ix, pr.l_bases, pr.l_quals, pr.offsets, pr.r_bases, pr.r_quals, pr.offsets, L)
    rec2, hb2, hq2, tot2 = b.download()
    assert tot == tot2 and rec.tobytes() == rec2.tobytes() and hb == hb2 and hq == hq2
    assert tot["overflow"] == 0 and tot["hits"] > 2000
    _, _, moff, _ = fast_merge_device(ix, pr.l_bases, pr.l_quals, pr.offsets, pr.r_bases, pr.r_quals, pr.offsets, L)
    assert int(((moff[1:] - moff[:-1]) > 0).sum()) == tot["merged_pairs"]
    key = rec["pair_id"].astype(np.int64) * 4 + rec["source"]
    assert (np.diff(key) > 0).all()
    ox = oracle.OracleIndexer(genes.seqs)
    for h in rec[:: max(1, rec.shape[0] // 3000)]:
        o, ln = int(h["seq_offset"]), int(h["read_len"])
        got = [(int(h["m"][k]["seq_start"]), int(h["m"][k]["seq_end"]), int(h["m"][k]["contig"]), int(h["m"][k]["position"]))
               for k in range(2)]
        assert ox.map_read(hb[o:o + ln]) == got
        assert oracle.in_required_direction(got, genes.reversed_flags)
    ix.close()


@pytest.mark.gpu
@pytest.mark.parametrize("chunk_bytes,final_newline", [(40_000, True), (7_000, False), (1_000_000, True)])
def test_text_stream_equals_the_one_shot_scan(gpu_device, chunk_bytes, final_newline):
    """FASTQ text handed over in raw chunks (boundaries anywhere: mid-line, mid-record, the two files out
    of step because R2's names are longer) gives the records of the one-shot scan of the whole files;
    R2 has three records more than R1 (the shorter file ends both)."""
    from genefuserust_amd import Indexer
    from genefuserust_amd.fastq import fastq_cut_device
    from genefuserust_amd.read_pair import scan_pairs_device
    from genefuserust_amd.scan_stream import scan_pair_text_stream
    import torch
    g = json.load(open(GOLDEN))
    genes = [None if x is None else x.encode() for x in g["genes"]]
    ix = Indexer.from_gene_slices(genes, g["reversed"])
    ix.make_index()
    rng = np.random.default_rng(3)
    pairs = _make_pairs(rng, genes, 700)
    t1 = b"\n".join(b"@p%d/1\n%s\n+\n%s" % (k, p[0], p[1]) for k, p in enumerate(pairs))
    extra = pairs + pairs[:3]
    t2 = b"\n".join(b"@pair_with_a_longer_name_%d/2\n%s\n+\n%s" % (k, p[2], p[3]) for k, p in enumerate(extra))
    if final_newline:
        t1, t2 = t1 + b"\n", t2 + b"\n"
    a1, a2 = np.frombuffer(t1, dtype=np.uint8).copy(), np.frombuffer(t2, dtype=np.uint8).copy()
    got = list(scan_pair_text_stream(ix, a1, a2, chunk_bytes=chunk_bytes, max_read_len=150))
    assert sum(t[3]["pairs"] for t in got) == 700
    assert len(got) >= (2 if chunk_bytes < 200_000 else 1)
    # the one-shot scan of the same records
    b1 = fastq_cut_device(ix, torch.from_numpy(a1).cuda())
    b2 = fastq_cut_device(ix, torch.from_numpy(a2).cuda())
    o2 = b2.offsets[:701]
    want = scan_pairs_device(ix, b1.bases, b1.quals, b1.offsets, b2.bases[:int(o2[-1])], b2.quals[:int(o2[-1])], o2, 150,
                             hits_cap=2100, bytes_cap=700_000).download()
    rec = np.concatenate([t[0] for t in got])
    assert rec.shape[0] == want[0].shape[0] > 100
    for f in ("pair_id", "source", "flags", "read_len", "merge_diff"):
        assert (rec[f] == want[0][f]).all(), f
    assert rec["m"].tobytes() == want[0]["m"].tobytes()
    # the reads travel with their records (offsets are per chunk)
    k = 0
    for r, hb, hq, tot in got:
        for h in r:
            o, ln = int(h["seq_offset"]), int(h["read_len"])
            w = want[0][k]
            wo = int(w["seq_offset"])
            assert hb[o:o + ln] == want[1][wo:wo + ln] and hq[o:o + ln] == want[2][wo:wo + ln]
            k += 1
    assert sum(t[3]["merged_pairs"] for t in got) == want[3]["merged_pairs"]
    ix.close()


@pytest.mark.gpu
def test_pack_sizes_beyond_the_32_bit_candidate_space_are_rejected(gpu_device):
    """gf_scan_pairs_device numbers its 3 n candidates in 32 bits: a larger pack is an error, not a wrap-around
    (the mapping entry splits large batches into spans instead: test_span_split_of_large_batches)."""
    import torch
    from genefuserust_amd import Indexer, _lib
    ix = Indexer.from_gene_slices([b"ACGT" * 100])
    ix.make_index()
    d = torch.zeros(64, dtype=torch.uint8, device="cuda")
    o = torch.zeros(8, dtype=torch.int64, device="cuda")
    tot = torch.zeros(8, dtype=torch.int64, device="cuda")
    rc = _lib.lib().gf_scan_pairs_device(ix._handle(), d.data_ptr(), d.data_ptr(), o.data_ptr(), 0, d.data_ptr(), d.data_ptr(),
                                         o.data_ptr(), 0, (1 << 31) // 3 + 1, 150, 0, 0, d.data_ptr(), 0, d.data_ptr(),
                                         d.data_ptr(), 0, tot.data_ptr(), 0)
    assert rc == _lib.GF_ERR_CAPACITY
    rc = _lib.lib().gf_scan_pairs_device(ix._handle(), d.data_ptr(), d.data_ptr(), o.data_ptr(), 0, d.data_ptr(), d.data_ptr(),
                                         o.data_ptr(), 0, 4, 3000, 0, 0, d.data_ptr(), 0, d.data_ptr(), d.data_ptr(), 0,
                                         tot.data_ptr(), 0)
    assert rc == _lib.GF_ERR_READ_TOO_LONG   # a merged read of 2 x 3000 bases would exceed GF_MAX_READ_LEN
    ix.close()


@pytest.mark.gpu
def test_scan_pairs_with_250_base_reads(gpu_device, oracle):
    """2 x 250-base pairs: merged reads of up to 470 bases leave the flat kernels' 320-base limit and take the
    1024-base wave-per-read class; R1 / R2 take the 16-word flat kernels.  Against the oracle-driven policy."""
    import torch
    from genefuserust_amd import FusionMapper, Indexer
    from genefuserust_amd.read_pair import finish_pair_hits, pack_reads, scan_pairs_device
    g = json.load(open(GOLDEN))
    genes = [None if x is None else x.encode() for x in g["genes"]]
    ix = Indexer.from_gene_slices(genes, g["reversed"])
    ix.make_index()
    ox = oracle.OracleIndexer(genes)
    rng = np.random.default_rng(12)
    g0, g1 = genes[0], genes[1]
    pairs = []
    for k in range(400):
        if k % 4 != 3:
            p, q = int(rng.integers(400, len(g0) - 400)), int(rng.integers(400, len(g1) - 500))
            frag = g0[p - 300:p] + g1[q:q + 300]
        else:
            frag = rand_seq(rng, 600)
        lo = int(rng.integers(0, 100))
        flen = int(rng.integers(260, 500))
        f = frag[lo:lo + flen]
        if k % 2:
            f = rc(f)
        rl = min(250, len(f))
        pairs.append((f[:rl], b"F" * rl, rc(f)[:rl], b"F" * rl))
    want, n_merged = _reference_policy(oracle, ox, g["reversed"], pairs)
    lb, lo_ = pack_reads([p[0] for p in pairs]); lq, _ = pack_reads([p[1] for p in pairs])
    rb, ro = pack_reads([p[2] for p in pairs]); rq, _ = pack_reads([p[3] for p in pairs])
    t = [torch.from_numpy(a).cuda() for a in (lb, lq, lo_, rb, rq, ro)]
    rec, hb, hq, tot = scan_pairs_device(ix, *t, 250, hits_cap=1200, bytes_cap=1200 * 500).download()
    assert tot["overflow"] == 0 and tot["merged_pairs"] == n_merged > 100
    flat = [(p, w) for p, ws in enumerate(want) for w in ws]
    assert tot["hits"] == len(flat) == rec.shape[0] > 100
    long_merged = 0
    for h, (p, (source, on_rc, m_rev, seq, qual, rm)) in zip(rec, flat):
        assert int(h["pair_id"]) == p and int(h["source"]) == source and bool(h["flags"] & 1) == on_rc
        o, ln = int(h["seq_offset"]), int(h["read_len"])
        assert hb[o:o + ln] == seq and hq[o:o + ln] == qual
        long_merged += source == 0 and ln > 320
    assert long_merged > 20
    done = finish_pair_hits(FusionMapper(ix), rec, hb, hq)
    for (pid, m), (p, (source, on_rc, m_rev, seq, qual, rm)) in zip(done, flat):
        assert (m.m_read_break, m.m_gap, m.m_left_distance, m.m_right_distance) == (
            rm["read_break"], rm["gap"], rm["left_distance"], rm["right_distance"])
    ix.close()
