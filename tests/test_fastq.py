"""FastqReader::read / FastqReaderPair::read (fastq_reader.rs:75-147,
:209-218) with the record cutting on the device.

CPU: the two restatements against each other on the reference's own test files
(testdata/R1.fq, R2.fq, kept as data under tests/golden/) and on texts that reach every
edge (no final newline, incomplete last record, empty lines, CR LF, empty text).
GPU: the device kernels behind the C ABI against the oracle, and the whole front of the
paired-end path — files -> records -> merge -> mapping — against the oracle run record by
record."""
import gzip
import json
import os

import numpy as np
import pytest

from oracle import indexer_model as M
from tests.helpers import rand_seq, rc

HERE = os.path.dirname(__file__)
R1 = os.path.join(HERE, "golden", "R1.fq")
R2 = os.path.join(HERE, "golden", "R2.fq")
GOLDEN = os.path.join(HERE, "golden", "branch_cases.json")

EDGE_TEXTS = [b"", b"\n", b"a", b"a\nb\nc\nd", b"a\nb\nc\nd\n", b"a\nb\nc\nd\ne\nf\ng", b"\n\n\n\n\n",
              b"a\r\nb\r\nc\r\nd\r\n", b"@r\nACGT\n+\nIIII\n@s\nAC\n+\nI\n", b"@r\nACGT\n+\nII\n@s\n\n+\n\n@t"]


def synth_fastq(seed: int, n: int, lens=(30, 160), ragged_quality: bool = False) -> bytes:
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        ln = int(rng.integers(*lens))
        seq = rand_seq(rng, ln)
        ql = ln if not ragged_quality or i % 7 else max(0, ln + int(rng.integers(-3, 4)))
        qual = bytes(rng.integers(35, 75, size=ql, dtype=np.uint8))
        name = b"@read%d %d:N:0:ACGT" % (i, seed) + b"x" * int(rng.integers(0, 40))
        out += [name, seq, b"+", qual]
    return b"\n".join(out) + (b"\n" if seed % 2 else b"")


def test_reference_test_files(oracle):
    """testdata/R1.fq / R2.fq: three records each, the last line without a newline."""
    for path, lens in ((R1, [151, 151, 151]), (R2, [151, 148, 148])):
        t = open(path, "rb").read()
        recs = oracle.fastq_cut(t)
        assert recs == M.fastq_records(t)
        assert [len(r[1]) for r in recs] == lens == [len(r[3]) for r in recs]
        assert all(r[0].startswith(b"@NB551106") and r[2] == b"+" for r in recs)


def test_oracle_and_model_agree_on_edges(oracle):
    for t in EDGE_TEXTS + [synth_fastq(s, 50) for s in (1, 2, 3)] + [synth_fastq(4, 40)[:-7]]:
        assert oracle.fastq_cut(t) == M.fastq_records(t), t[:60]


def test_extension_rules():
    from genefuserust_amd.fastq import FastqReader
    assert FastqReader("x.fq.gz").m_zipped and FastqReader("x.fastq.gz").m_zipped and FastqReader("x.fa.gz").m_zipped
    assert not FastqReader("x.fq").m_zipped and not FastqReader("x.fastq").m_zipped
    with pytest.raises(ValueError):
        FastqReader("x.txt")


def _cut(ix, text):
    import torch
    from genefuserust_amd.fastq import fastq_cut_device
    d = torch.from_numpy(np.frombuffer(text, dtype=np.uint8).copy()).cuda() if text else \
        torch.empty(0, dtype=torch.uint8, device="cuda")
    b = fastq_cut_device(ix, d)
    torch.cuda.synchronize()
    off = b.offsets.cpu().numpy()
    bb, qq = b.bases.cpu().numpy().tobytes(), b.quals.cpu().numpy().tobytes()
    seqs = [bb[off[i]:off[i + 1]] for i in range(b.n_records)]
    quals = [qq[off[i]:off[i + 1]] for i in range(b.n_records)]
    return b, seqs, quals


@pytest.fixture(scope="module")
def small_index(gpu_device):
    from genefuserust_amd import Indexer
    g = json.load(open(GOLDEN))
    ix = Indexer.from_gene_slices([None if x is None else x.encode() for x in g["genes"]], g["reversed"])
    ix.make_index()
    yield ix
    ix.close()


@pytest.mark.gpu
def test_fastq_cut_device_parity(small_index, oracle):
    from genefuserust_amd.fastq import record_lines
    texts = EDGE_TEXTS + [open(R1, "rb").read(), open(R2, "rb").read(), synth_fastq(5, 3000), synth_fastq(6, 9000),
                          synth_fastq(7, 700, lens=(0, 400)), synth_fastq(8, 300)[:-11],
                          b"\n" * 70000, synth_fastq(9, 20000, lens=(140, 152))]
    for t in texts:
        want = oracle.fastq_cut(t)
        b, seqs, quals = _cut(small_index, t)
        assert b.n_records == len(want), t[:40]
        assert seqs == [w[1] for w in want]
        # a quality line is cut or padded with '!' to its sequence's length (include/gfmatch.h)
        assert quals == [(w[3][:len(w[1])] + b"!" * max(0, len(w[1]) - len(w[3]))) for w in want]
        assert b.n_bad_quality == sum(len(w[1]) != len(w[3]) for w in want)
        for i in sorted({0, 1, len(want) // 2, len(want) - 1} & set(range(len(want)))):
            assert record_lines(b, t, i) == want[i]


@pytest.mark.gpu
def test_quality_of_a_different_length(small_index, oracle):
    t = synth_fastq(10, 500, ragged_quality=True)
    want = oracle.fastq_cut(t)
    b, seqs, quals = _cut(small_index, t)
    assert seqs == [w[1] for w in want]
    bad = 0
    for w, q in zip(want, quals):
        ln = len(w[1])
        assert q == (w[3][:ln] + b"!" * max(0, ln - len(w[3])))
        bad += len(w[3]) != ln
    assert b.n_bad_quality == bad > 0


@pytest.mark.gpu
def test_files_to_matches(small_index, oracle, tmp_path):
    """FASTQ files (one plain, one gzipped) -> device records -> fast_merge -> mapping, against
    the oracle fed record by record.  Pairs are cut from planted fusions of the golden genes."""
    import torch
    from genefuserust_amd.fastq import FastqReaderPair
    from genefuserust_amd.read_pair import fast_merge_device
    g = json.load(open(GOLDEN))
    genes = [None if x is None else x.encode() for x in g["genes"]]
    rng = np.random.default_rng(77)
    g0, g1 = genes[0], genes[1]
    l_txt, r_txt = [], []
    for k in range(400):
        p, q = int(rng.integers(300, 2600)), int(rng.integers(300, 2200))
        frag = (g0[p - 130:p] + g1[q:q + 130]) if k % 2 else rand_seq(rng, 260)
        f = frag[int(rng.integers(0, 30)):][:int(rng.integers(170, 240))]
        ln1, ln2 = int(rng.integers(140, 152)), int(rng.integers(140, 152))
        s1, s2 = f[:ln1], rc(f)[:ln2]
        l_txt += [b"@p%d/1" % k, s1, b"+", b"F" * len(s1)]
        r_txt += [b"@p%d/2" % k, s2, b"+", b"F" * len(s2)]
    r_txt += [b"@extra/2", b"ACGT", b"+", b"FFFF"]   # the longer file's tail is ignored (fastq_reader.rs:213)
    lp, rp = tmp_path / "L.fq", tmp_path / "R.fastq.gz"
    lp.write_bytes(b"\n".join(l_txt) + b"\n")
    with gzip.open(rp, "wb") as f:
        f.write(b"\n".join(r_txt))
    (l, lt), (r, rtxt) = FastqReaderPair.from_paths(str(lp), str(rp)).read_all_device(small_index)
    assert l.n_records == r.n_records == 400
    mx = max(l.max_read_len(), r.max_read_len())
    bases, quals, off, diff = fast_merge_device(small_index, l.bases, l.quals, l.offsets, r.bases, r.quals, r.offsets, mx)
    lens = off[1:] - off[:-1]
    counts, _ = small_index.map_reads_device(bases, off, int(lens.max()))
    torch.cuda.synchronize()
    cn = counts.cpu().numpy()
    ox = oracle.OracleIndexer(genes)
    lrec, rrec = oracle.fastq_cut(lt), oracle.fastq_cut(rtxt)
    n_hit = n_merged = 0
    for k in range(400):
        m = oracle.fast_merge(lrec[k][1], lrec[k][3], rrec[k][1], rrec[k][3])
        exp = 0 if m is None else len(ox.map_read(m[0]))
        assert int(cn[k]) == exp, k
        n_merged += m is not None
        n_hit += exp > 0
    assert n_merged > 200 and n_hit > 50


@pytest.mark.gpu
def test_fastq_cut_full_size_round_trip(small_index):
    """BASELINE-size property: 5 M fixed-length records (1.6 GB of text) written on the device,
    cut by the kernels, and compared with the tensors they were written from — every base and
    every quality, no oracle involved."""
    import torch
    from genefuserust_amd.fastq import fastq_cut_device
    from tools.bench_frontend import make_text
    n, L = 5_000_000, 150
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    acgt = torch.tensor(list(b"ACGTN"), dtype=torch.uint8, device=dev)
    bases = acgt[torch.randint(0, 5, (n * L,), generator=g, device=dev)]
    quals = (33 + torch.randint(0, 42, (n * L,), generator=g, device=dev)).to(torch.uint8)
    text = make_text(bases, quals, n, L, 1, dev)
    batch = fastq_cut_device(small_index, text)
    assert batch.n_records == n and batch.n_bad_quality == 0
    assert torch.equal(batch.offsets, torch.arange(0, (n + 1) * L, L, dtype=torch.int64, device=dev))
    assert torch.equal(batch.bases, bases) and torch.equal(batch.quals, quals)
    # without the final newline the last record still counts; without its last line it does not
    assert fastq_cut_device(small_index, text[:-1]).n_records == n
    assert fastq_cut_device(small_index, text[:-(L + 1)]).n_records == n - 1


@pytest.mark.gpu
def test_gather_through_the_abi_with_awkward_buffers(small_index, oracle):
    """gf_fastq_gather_device called directly: output buffers that are not 16-byte aligned (the
    ragged copy's pieces cannot be stored whole: every tile takes the wavefront-per-record
    path), a text that starts at an odd address, and output buffers too small for all records
    (the records that fit are copied, `offsets` still tells how much is needed)."""
    import torch
    from genefuserust_amd import _lib
    L, h = _lib.lib(), small_index._handle()
    t = synth_fastq(11, 4000, lens=(30, 200))
    want = oracle.fastq_cut(t)
    dev = torch.device("cuda", small_index.info()["device"])
    st = torch.cuda.current_stream(dev).cuda_stream
    for text_shift, out_shift, cap_cut in ((0, 1, 0), (3, 0, 0), (1, 5, 0), (0, 0, 50000), (0, 8, 50000)):
        raw = torch.zeros(len(t) + 64, dtype=torch.uint8, device=dev)
        text = raw[text_shift:text_shift + len(t)]
        text.copy_(torch.frombuffer(bytearray(t), dtype=torch.uint8))
        n = len(t)
        ws = torch.empty(int(L.gf_fastq_workspace_bytes(n)), dtype=torch.uint8, device=dev)
        n_lines = torch.zeros(2, dtype=torch.int64, device=dev)
        nl_pos = torch.empty(n, dtype=torch.int64, device=dev)
        _lib.check(L.gf_fastq_index_device(h, text.data_ptr(), n, nl_pos.data_ptr(), n, n_lines.data_ptr(),
                                           ws.data_ptr(), st))
        lines, newlines = (int(x) for x in n_lines.cpu())
        n_rec = lines // 4
        assert n_rec == len(want)
        total = sum(len(w[1]) for w in want)
        cap = total - cap_cut
        offsets = torch.zeros(n_rec + 1, dtype=torch.int64, device=dev)
        bases_raw = torch.full((total + 64,), 0x2E, dtype=torch.uint8, device=dev)
        quals_raw = torch.full((total + 64,), 0x2E, dtype=torch.uint8, device=dev)
        bases, quals = bases_raw[out_shift:], quals_raw[out_shift:]
        n_bad = torch.zeros(1, dtype=torch.int64, device=dev)
        _lib.check(L.gf_fastq_gather_device(h, text.data_ptr(), n, nl_pos.data_ptr(), newlines, n_rec,
                                            offsets.data_ptr(), bases.data_ptr(), quals.data_ptr(), cap,
                                            n_bad.data_ptr(), ws.data_ptr(), st))
        torch.cuda.synchronize(dev)
        off = offsets.cpu().numpy()
        assert off[-1] == total and int(n_bad.item()) == 0
        b, q = bases.cpu().numpy().tobytes(), quals.cpu().numpy().tobytes()
        copied = 0
        for i, w in enumerate(want):
            lo, hi = int(off[i]), int(off[i + 1])
            if hi <= cap:
                assert b[lo:hi] == w[1] and q[lo:hi] == w[3], (text_shift, out_shift, cap_cut, i)
                copied += 1
        assert copied == len(want) if cap_cut == 0 else 0 < copied < len(want)
        # nothing beyond the capacity handed over is written
        assert b[cap:cap + 32] == b"." * 32 and q[cap:cap + 32] == b"." * 32
        if out_shift:
            assert bases_raw[:out_shift].cpu().numpy().tobytes() == b"." * out_shift
