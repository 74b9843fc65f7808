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


@pytest.mark.gpu
def test_tail_on_the_device_equals_the_host_tail(gpu_device, oracle):
    """gf_pair_hits_finish_device (a wavefront per record and side, bit-vector Levenshtein with ballots) against
    gf_pair_hits_finish (host) and the oracle's calc_distance, on crafted records: both strands, sides that cross a
    strand (-1) or leave the gene (-2), lengths around the 64-symbol block edges and beyond ten blocks, reads with N and
    lower case, substitutions and indels against the gene."""
    import torch
    from genefuserust_amd import Indexer, _lib
    from genefuserust_amd.read_pair import PairScan, finish_pair_hits_device
    from tests.helpers import rand_seq, rc
    rng = np.random.default_rng(77)
    genes = [rand_seq(rng, 9000), rand_seq(rng, 7000), rand_seq(rng, 3000)]
    g1 = bytearray(genes[1]); g1[2000] = ord("N"); genes[1] = bytes(g1)
    ix = Indexer.from_gene_slices(genes, [False, True, False])
    ix.make_index()
    recs, blob = [], bytearray()

    def mutate(s: bytes) -> bytes:
        s = bytearray(s)
        for _ in range(int(rng.integers(0, 4))):
            k = int(rng.integers(0, max(len(s), 1)))
            op = int(rng.integers(0, 3))
            if op == 0 and s:
                s[k] = b"ACGTNa"[int(rng.integers(0, 6))]
            elif op == 1 and len(s) > 2:
                del s[k]
            else:
                s.insert(k, b"ACGT"[int(rng.integers(0, 4))])
        return bytes(s)

    lens = [2, 40, 63, 64, 65, 127, 128, 129, 150, 270, 639, 640, 641, 700, 1300]
    for t in range(260):
        la, lb = int(rng.choice(lens)), int(rng.choice(lens))
        ca, cb = int(rng.integers(0, 3)), int(rng.integers(0, 3))
        pa = int(rng.integers(la + 5, len(genes[ca]) - la - 5))
        pb = int(rng.integers(lb + 5, len(genes[cb]) - lb - 5))
        left = genes[ca][pa - la + 1:pa + 1]
        right = genes[cb][pb:pb + lb]
        sa, sb = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        # a read = left part | right part; a part on the reverse strand carries a negative start position
        lpart = mutate(rc(left) if sa else left)
        rpart = mutate(rc(right) if sb else right)
        read = lpart + rpart
        ll = len(lpart)
        lpos = -(pa) if sa else pa - ll + 1           # start_gp of the left segment (read base 0)
        rpos = -(pb + lb - 1) - ll if sb else pb - ll   # start_gp of the right segment, also counted from read base 0
        if t % 17 == 0:
            lpos = 3 - ll                               # crosses position 0: -1
        if t % 19 == 0:
            rpos = len(genes[cb]) + 50                  # beyond the gene: -2
        m0 = (0, ll - 1, lpos, ca)
        m1 = (ll, len(read) - 1, rpos, cb)
        if t % 2:
            m0, m1 = m1, m0                             # TOP / SECOND in either order
        recs.append((t, 1, 0, len(read), 0, len(blob), m0, m1))
        blob += read
    n = len(recs)
    rec = np.zeros(n, dtype=_lib.PAIR_HIT_DTYPE)
    for k, (pid, src, fl, ln, md, off, m0, m1) in enumerate(recs):
        rec[k]["pair_id"], rec[k]["source"], rec[k]["flags"], rec[k]["read_len"] = pid, src, fl, ln
        rec[k]["merge_diff"], rec[k]["seq_offset"] = md, off
        for j, m in enumerate((m0, m1)):
            rec[k]["m"][j]["seq_start"], rec[k]["m"][j]["seq_end"], rec[k]["m"][j]["position"], rec[k]["m"][j]["contig"] = m
    hb = bytes(blob)
    want = np.zeros(n, dtype=_lib.READMATCH_DTYPE)
    st = np.zeros(n, dtype=np.int32)
    _lib.check(_lib.lib().gf_pair_hits_finish(ix._handle(), rec.ctypes.data, n, hb, len(hb), want.ctypes.data, st.ctypes.data, 4))
    assert (st == _lib.GF_RM_MATCH).all()
    assert (want["left_distance"] == -1).sum() + (want["right_distance"] == -1).sum() > 5
    assert (want["left_distance"] == -2).sum() + (want["right_distance"] == -2).sum() > 5
    assert (want["left_distance"] > 0).sum() > 50 and (want["left_distance"] == 0).sum() > 10
    dev = torch.device("cuda", gpu_device)
    cap = n + 7
    d_hits = torch.zeros((cap, 64), dtype=torch.uint8, device=dev)
    d_hits[:n] = torch.from_numpy(rec.view(np.uint8).reshape(n, 64)).to(dev)
    totals = torch.zeros(8, dtype=torch.int64, device=dev)
    totals[0] = n
    d_b = torch.from_numpy(np.frombuffer(hb, dtype=np.uint8).copy()).to(dev)
    out, status = finish_pair_hits_device(ix, PairScan(d_hits, d_b, d_b, totals))
    got = out[:n].cpu().numpy().view(_lib.READMATCH_DTYPE).reshape(-1)
    assert (status[:n].cpu().numpy() == _lib.GF_RM_MATCH).all() and (status[n:].cpu().numpy() == 0).all()
    for f in _lib.READMATCH_DTYPE.names:
        bad = np.nonzero(got[f] != want[f])[0]
        assert bad.size == 0, (f, bad[:5], got[f][bad[:5]], want[f][bad[:5]], [recs[i][3] for i in bad[:5]])
    # the oracle's own distances on a sample (calc_distance through orc_fusion_map_read needs the direction gate: use
    # plain edit distances of the forward-strand parts instead)
    for k in range(0, n, 9):
        h, w = rec[k], want[k]
        if w["left_distance"] >= 0 and w["left_position"] - (w["read_break"] + 1) + 1 >= 0:
            seq = hb[int(h["seq_offset"]):int(h["seq_offset"]) + int(w["read_break"]) + 1]
            g = genes[int(w["left_contig"])].upper()
            s0 = int(w["left_position"]) - len(seq) + 1
            assert oracle.edit_distance(seq, g[s0:s0 + len(seq)]) == int(w["left_distance"])
    # a record that names a gene the index does not have: flagged, the others untouched
    rec2 = rec[:3].copy()
    rec2[1]["m"][0]["contig"] = 9
    d_hits[:3] = torch.from_numpy(rec2.view(np.uint8).reshape(3, 64)).to(dev)
    totals[0] = 3
    out, status = finish_pair_hits_device(ix, PairScan(d_hits, d_b, d_b, totals))
    s3 = status[:3].cpu().numpy()
    assert s3[0] == _lib.GF_RM_MATCH and s3[1] == _lib.GF_ERR_ARG and s3[2] == _lib.GF_RM_MATCH
    ix.close()


@pytest.mark.gpu
def test_tail_on_the_device_after_a_real_pair_scan(gpu_device):
    """files-shaped flow: gf_scan_pairs_device then the device tail on its records, against the host tail."""
    import torch
    from genefuserust_amd import Indexer, _lib, synth
    from genefuserust_amd.read_pair import finish_pair_hits_device, scan_pairs_device
    genes = synth.make_geneset("IDX-T", scale=0.1)
    ix = Indexer.from_gene_slices(genes.seqs, genes.reversed_flags)
    ix.make_index()
    synth.MIXES["JUNC"] = (0.1, 0.4, 0.5)
    pr = synth.make_pairs(genes, 60_000, read_len=150, mix="JUNC", seed=4, device="cuda")
    scan = scan_pairs_device(ix, pr.l_bases, pr.l_quals, pr.offsets, pr.r_bases, pr.r_quals, pr.offsets, 150,
                             hits_cap=60_000, bytes_cap=60_000 * 320)
    out, status = finish_pair_hits_device(ix, scan)
    rec, hb, hq, tot = scan.download()
    n = rec.shape[0]
    assert n > 500 and tot["overflow"] == 0
    want = np.zeros(n, dtype=_lib.READMATCH_DTYPE)
    st = np.zeros(n, dtype=np.int32)
    recc = np.ascontiguousarray(rec)
    _lib.check(_lib.lib().gf_pair_hits_finish(ix._handle(), recc.ctypes.data, n, hb, len(hb), want.ctypes.data, st.ctypes.data, 4))
    got = out[:n].cpu().numpy().view(_lib.READMATCH_DTYPE).reshape(-1)
    assert got.tobytes() == want.tobytes() and (status[:n].cpu().numpy() == st).all()
    ix.close()


@pytest.mark.gpu
def test_qualities_left_in_the_text_give_the_same_scan(gpu_device, oracle):
    """fastq_cut_device(lean=True) + scan_pairs_device(l_qual_off=..): the qualities never leave the FASTQ text (the
    pipeline reads them at an overlap's mismatching columns, for reverse-complement retries and for hit records
    only).  Hit records, their bases AND their qualities equal the full gather's byte for byte — on pairs whose
    qualities matter: random Phred values, so that the merge rule's >= Q30 / <= Q15 test (read.rs:380-428) decides
    overlaps both ways and merged qualities differ from either read's."""
    import torch
    from genefuserust_amd import Indexer, synth
    from genefuserust_amd.fastq import fastq_cut_device
    from genefuserust_amd.read_pair import scan_pairs_device
    from tools.bench_frontend import make_text
    dev = torch.device("cuda")
    genes = synth.make_geneset("IDX-T", scale=0.1)
    ix = Indexer.from_gene_slices(genes.seqs, genes.reversed_flags)
    ix.make_index()
    synth.MIXES["JUNC"] = (0.1, 0.4, 0.5)
    n, L = 80_000, 150
    pr = synth.make_pairs(genes, n, read_len=L, mix="JUNC", seed=11, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(5)
    lq = (torch.randint(0, 42, pr.l_quals.shape, generator=g, device=dev) + 33).to(torch.uint8)
    rq = (torch.randint(0, 42, pr.r_quals.shape, generator=g, device=dev) + 33).to(torch.uint8)
    # a few mismatches inside the overlaps, so that the quality rule is actually consulted
    rb = pr.r_bases.clone()
    pos = torch.randint(0, rb.numel(), (n // 2,), generator=g, device=dev)
    rb[pos] = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[torch.randint(0, 4, (n // 2,), generator=g, device=dev)]
    t1 = make_text(pr.l_bases, lq, n, L, 1, dev)
    t2 = make_text(rb, rq, n, L, 2, dev)
    full1, full2 = fastq_cut_device(ix, t1), fastq_cut_device(ix, t2)
    lean1, lean2 = fastq_cut_device(ix, t1, lean=True), fastq_cut_device(ix, t2, lean=True)
    assert lean1.qual_off is not None and lean2.qual_off is not None and full1.qual_off is None
    assert torch.equal(lean1.bases, full1.bases) and torch.equal(lean1.offsets, full1.offsets)
    # qual_off points at the quality lines: the same bytes as the gathered qualities
    k = 1000
    qo, off = lean2.qual_off[:k].cpu().numpy(), full2.offsets[:k + 1].cpu().numpy()
    text2, q2 = t2.cpu().numpy().tobytes(), full2.quals.cpu().numpy().tobytes()
    assert all(text2[qo[i]:qo[i] + (off[i + 1] - off[i])] == q2[off[i]:off[i + 1]] for i in range(k))
    caps = dict(hits_cap=n, bytes_cap=n * 2 * L)
    a = scan_pairs_device(ix, full1.bases, full1.quals, full1.offsets, full2.bases, full2.quals, full2.offsets, L, **caps)
    b = scan_pairs_device(ix, lean1.bases, lean1.quals, lean1.offsets, lean2.bases, lean2.quals, lean2.offsets, L,
                          l_qual_off=lean1.qual_off, r_qual_off=lean2.qual_off, **caps)
    ra, ba, qa, ta = a.download()
    rb_, bb, qb, tb = b.download()
    assert ta == tb and ta["overflow"] == 0 and ta["hits"] > 500 and ta["merged_pairs"] > 1000 and ta["retried_reads"] > 0
    assert ra.tobytes() == rb_.tobytes() and ba == bb and qa == qb
    # and the merged reads among the hits carry merged qualities (neither read's own)
    assert any(int(r["source"]) == 0 for r in ra)
    # a text with a quality line of another length: the lean cut steps back to the full one
    bad = b"@r\nACGTACGTACGTACGTACGT\n+\nIIII\n"
    tb_ = torch.from_numpy(np.frombuffer(bad, dtype=np.uint8).copy()).to(dev)
    fb = fastq_cut_device(ix, tb_, lean=True)
    assert fb.qual_off is None and fb.n_bad_quality == 1 and fb.quals.cpu().numpy().tobytes() == b"IIII" + b"!" * 16
    ix.close()


@pytest.mark.gpu
@pytest.mark.parametrize("max_len", [150, 250, 300])
def test_qualities_in_the_text_with_ragged_reads(gpu_device, max_len):
    """The lean path again on texts written record by record: reads of 40 .. max_len bases (the 10-word, the 16-word
    and the byte-loop merge search, by max_len), qualities of their own per base, overlapping fragments with planted
    mismatches, a last record without its newline — against the full gather + gf_scan_pairs_device."""
    import torch
    from genefuserust_amd import Indexer
    from genefuserust_amd.fastq import fastq_cut_device
    from genefuserust_amd.read_pair import scan_pairs_device
    g = json.load(open(GOLDEN))
    genes = [None if x is None else x.encode() for x in g["genes"]]
    ix = Indexer.from_gene_slices(genes, g["reversed"])
    ix.make_index()
    rng = np.random.default_rng(1000 + max_len)
    g0, g1 = genes[0], genes[1]
    t1, t2 = [], []
    n = 1500
    for k in range(n):
        if k % 3 != 2:
            p, q = int(rng.integers(400, len(g0) - 400)), int(rng.integers(400, len(g1) - 500))
            frag = g0[p - 300:p] + g1[q:q + 300]
        else:
            frag = rand_seq(rng, 600)
        lo = int(rng.integers(0, 100))
        flen = int(rng.integers(60, 2 * max_len))
        f = bytearray(frag[lo:lo + flen])
        l1, l2 = int(rng.integers(40, max_len + 1)), int(rng.integers(40, max_len + 1))
        r1 = bytes(f[:l1])
        r2 = bytearray(rc(bytes(f))[:l2])
        for _ in range(int(rng.integers(0, 3))):          # mismatches between the mates
            r2[int(rng.integers(0, len(r2)))] = ord("ACGT"[int(rng.integers(0, 4))])
        q1 = bytes(rng.integers(33, 75, size=len(r1), dtype=np.uint8))
        q2 = bytes(rng.integers(33, 75, size=len(r2), dtype=np.uint8))
        t1.append(b"@p%d/1\n" % k + r1 + b"\n+\n" + q1 + b"\n")
        t2.append(b"@p%d/2\n" % k + bytes(r2) + b"\n+\n" + q2 + b"\n")
    text1, text2 = b"".join(t1)[:-1], b"".join(t2)      # (R1's last line ends with the file)
    d1 = torch.from_numpy(np.frombuffer(text1, dtype=np.uint8).copy()).cuda()
    d2 = torch.from_numpy(np.frombuffer(text2, dtype=np.uint8).copy()).cuda()
    f1, f2 = fastq_cut_device(ix, d1), fastq_cut_device(ix, d2)
    l1_, l2_ = fastq_cut_device(ix, d1, lean=True), fastq_cut_device(ix, d2, lean=True)
    assert f1.n_records == f2.n_records == l1_.n_records == n and l1_.qual_off is not None
    caps = dict(hits_cap=3 * n, bytes_cap=3 * n * 2 * max_len)
    a = scan_pairs_device(ix, f1.bases, f1.quals, f1.offsets, f2.bases, f2.quals, f2.offsets, max_len, **caps)
    b = scan_pairs_device(ix, l1_.bases, l1_.quals, l1_.offsets, l2_.bases, l2_.quals, l2_.offsets, max_len,
                          l_qual_off=l1_.qual_off, r_qual_off=l2_.qual_off, **caps)
    ra, ba, qa, ta = a.download()
    rb_, bb, qb, tb = b.download()
    assert ta == tb and ta["overflow"] == 0 and ta["hits"] >= 10 and ta["merged_pairs"] > 100 and ta["retried_reads"] >= 0
    assert ra.tobytes() == rb_.tobytes() and ba == bb and qa == qb
    ix.close()
