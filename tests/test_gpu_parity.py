"""Parity of the HIP path (through the C ABI) with the oracle — needs an MI355X.

Bit-exact bar: identical SeqMatch lists (values and order) for every read,
identical index content for every k-mer.  Nothing here reads /root/reference.
"""
import json
import os

import numpy as np
import pytest

from tests.helpers import branch_genes, branch_reads, matches_to_tuples, rand_seq, rc

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "branch_cases.json")


@pytest.fixture(scope="module")
def golden():
    return json.load(open(GOLDEN))


@pytest.fixture(scope="module")
def branch_index(gpu_device, golden):
    from genefuserust_amd import Indexer
    genes = [None if x is None else x.encode() for x in golden["genes"]]
    ix = Indexer.from_gene_slices(genes, golden["reversed"])
    ix.make_index()
    yield ix
    ix.close()


def _compare_reads(ix, ox, reads, label=""):
    from genefuserust_amd.synth import ragged_batch
    bases, offsets = ragged_batch(reads)
    counts, matches = ix.map_reads_packed(bases, offsets)
    ocounts, omatches = ox.map_reads_packed(bases, offsets, threads=8)
    got = matches_to_tuples(counts, matches)
    want = matches_to_tuples(ocounts, omatches)
    bad = [r for r in range(len(reads)) if got[r] != want[r]]
    assert not bad, "%s: %d/%d reads differ, first %d: got %s want %s read %r" % (
        label, len(bad), len(reads), bad[0], got[bad[0]], want[bad[0]], reads[bad[0]])
    return sum(1 for w in want if w)


def test_index_stats_and_fusion_seq(branch_index, golden, oracle):
    info = branch_index.info()
    genes = [None if x is None else x.encode() for x in golden["genes"]]
    assert info["n_genes"] == len(genes)
    assert info["n_keys"] == golden["stats"]["n_keys"]
    assert info["n_high_keys"] == golden["stats"]["n_high_keys"]
    assert info["n_unique"] == golden["stats"]["m_unique_pos"]
    # m_dupe_pos counts keys that ever became NORMAL dupes, HIGH ones included
    assert info["n_dupe_keys"] + info["n_high_keys"] == golden["stats"]["m_dupe_pos"]
    ox = oracle.OracleIndexer(genes)
    assert branch_index.m_fusion_seq == [ox.fusion_seq(c) for c in range(len(genes))]
    assert branch_index.m_fusion_seq[2] == golden["genes"][2].upper()
    assert branch_index.m_fusion_seq[3] == ""


def test_index_content_every_key(branch_index, golden):
    keys = np.array([k for k, _, _ in golden["index"]], dtype=np.uint32)
    cnt, ctg, pos = branch_index.lookup(keys)
    for j, (k, n, sites) in enumerate(golden["index"]):
        assert cnt[j] == n, (k, cnt[j], n)
        if n > 0:
            assert [(int(ctg[j, t]), int(pos[j, t])) for t in range(n)] == [tuple(s) for s in sites], k
    # absent k-mers stay absent (exact membership, like the reference's 2^32-bit bitmap)
    rng = np.random.default_rng(5)
    present = set(int(k) for k in keys)
    absent = np.array([k for k in rng.integers(0, 1 << 32, size=20000, dtype=np.uint64) if int(k) not in present],
                      dtype=np.uint32)
    cnt, _, _ = branch_index.lookup(absent)
    assert not cnt.any()
    # neighbours of present keys (one base changed) are the likeliest false positives
    near = np.array([k ^ (1 << int(b)) for k in list(present)[:3000] for b in (0, 7, 31)], dtype=np.uint32)
    near = np.array([k for k in near if int(k) not in present], dtype=np.uint32)
    cnt, _, _ = branch_index.lookup(near)
    assert not cnt.any()


@pytest.mark.parametrize("variant", [0, "0-pack", 1, 2])
def test_map_golden_cases(branch_index, golden, variant):
    """variant 0 = flat pipeline (pack fused into seed+verify; the host call's zero-copy route switched off),
    "0-pack" = the default host call: a pack of this size takes the zero-copy route (one launch of the wave-per-read
    kernel over pinned memory), 1 = wave-per-read probe-all, 2 = wave-per-read seed+verify."""
    branch_index.set_pack_call_reads(-1 if variant == "0-pack" else 0)
    branch_index.set_map_variant(0 if variant == "0-pack" else variant)
    reads = [c["read"].encode() for c in golden["cases"]]
    got = branch_index.map_reads(reads)
    branch_index.set_map_variant(0)
    branch_index.set_pack_call_reads(-1)
    for c, g in zip(golden["cases"], got):
        flat = [(m.seq_start, m.seq_end, m.start_gp.contig, m.start_gp.position) for m in g]
        assert flat == [tuple(m) for m in c["expect"]], c["label"]


def test_map_read_single_call(branch_index, golden):
    from genefuserust_amd import GenePos, SeqMatch
    for c in golden["cases"][:48]:
        got = branch_index.map_read(c["read"])
        assert got == [SeqMatch(a, b, GenePos(ct, p)) for a, b, ct, p in c["expect"]], c["label"]


def test_in_required_direction_on_golden(branch_index, golden, oracle):
    reads = [c["read"].encode() for c in golden["cases"]]
    for c, g in zip(golden["cases"], branch_index.map_reads(reads)):
        want = oracle.in_required_direction([tuple(m) for m in c["expect"]], golden["reversed"])
        assert branch_index.in_required_direction(g) == want, c["label"]


def test_kernel_variants_and_alignment(branch_index, golden):
    """The three LDS footprints (<=256, <=1024, <=4096) and every byte alignment of
    the read start give the same answer."""
    import torch
    from genefuserust_amd.synth import ragged_batch
    reads = [c["read"].encode() for c in golden["cases"]]
    want = [[tuple(m) for m in c["expect"]] for c in golden["cases"]]
    for pad in (0, 1, 2, 3):
        bases, offsets = ragged_batch(reads)
        bases = np.concatenate([np.frombuffer(b"G" * pad, dtype=np.uint8), bases])
        offsets = offsets + pad
        d_b = torch.from_numpy(bases).cuda()
        d_o = torch.from_numpy(offsets).cuda()
        for lcap in (270, 1024, 4096):
            counts, matches = branch_index.map_reads_device(d_b, d_o, lcap)
            torch.cuda.synchronize()
            c = counts.cpu().numpy().astype(np.int32)[:len(reads)]
            m = matches.cpu().numpy().view(np.dtype([("seq_start", "<i4"), ("seq_end", "<i4"),
                                                    ("position", "<i4"), ("contig", "<i2"), ("pad", "<i2")]))
            m = m.reshape(-1, 2)[:len(reads)]
            assert matches_to_tuples(c, m) == want, (pad, lcap)
    # a batch limit below the longest read marks that read instead of mapping it
    bases, offsets = ragged_batch(reads)
    counts, _ = branch_index.map_reads_device(torch.from_numpy(bases).cuda(), torch.from_numpy(offsets).cuda(), 256)
    c = counts.cpu().numpy()
    lens = np.diff(offsets)
    assert (c[lens > 256] == 255).all() and (c[lens <= 256] <= 2).all()


def test_hits_compaction_is_ordered_and_complete(branch_index, golden):
    from genefuserust_amd.synth import ragged_batch
    reads = [c["read"].encode() for c in golden["cases"]] * 40  # > 2 compaction tiles
    bases, offsets = ragged_batch(reads)
    counts, matches = branch_index.map_reads_packed(bases, offsets)
    hits = branch_index.map_reads_hits(bases, offsets, read_id_base=1000)
    idx = np.nonzero(counts)[0]
    assert hits["read_id"].tolist() == (idx + 1000).tolist()
    assert hits["n"].tolist() == counts[idx].tolist()
    for h, r in zip(hits, idx):
        for k in range(int(h["n"])):
            assert h["m"][k] == matches[r, k]
    # capacity smaller than the number of hits: total still reported, prefix written
    few = branch_index.map_reads_hits(bases, offsets, cap=5)
    assert few["read_id"].tolist() == idx[:5].tolist()


def test_edge_batches(branch_index, gpu_device, oracle):
    from genefuserust_amd import Indexer, _lib
    # empty batch
    counts, matches = branch_index.map_reads_packed(np.zeros(0, np.uint8), np.zeros(1, np.int64))
    assert counts.size == 0
    # batch of only empty / tiny reads
    assert branch_index.map_reads([b"", b"A", b"ACGT" * 4, b""]) == [[], [], [], []]
    # too long
    with pytest.raises(_lib.GfError) as e:
        branch_index.map_read(b"A" * 5000)
    assert e.value.code == _lib.GF_ERR_READ_TOO_LONG
    with pytest.raises(_lib.GfError):
        branch_index.map_read(b"A" * (_lib.GF_MAX_READ_LEN + 1))
    # maximum size: a read of exactly GF_MAX_READ_LEN bases stitched from two genes
    g = [x for x in branch_index.m_fusion_seq[:2]]
    big = (g[0][100:2100] + g[1][200:2296]).encode()
    assert len(big) == _lib.GF_MAX_READ_LEN
    ox_genes = [s.encode() if s else None for s in branch_index.m_fusion_seq]
    want = oracle.OracleIndexer(ox_genes).map_read(big)
    got = [(m.seq_start, m.seq_end, m.start_gp.contig, m.start_gp.position) for m in branch_index.map_read(big)]
    assert got == want
    # the same on repeat-free genes, where the maximum-size read really yields two segments
    rng = np.random.default_rng(77)
    clean = [rand_seq(rng, 3000), rand_seq(rng, 3000)]
    ixc = Indexer.from_gene_slices(clean)
    ixc.make_index()
    big = clean[0][10:2058] + clean[1][500:2548]
    assert len(big) == _lib.GF_MAX_READ_LEN
    want = oracle.OracleIndexer(clean).map_read(big)
    got = [(m.seq_start, m.seq_end, m.start_gp.contig, m.start_gp.position) for m in ixc.map_read(big)]
    assert got == want and len(want) == 2
    ixc.close()
    # BASELINE config 1 plumbing: every gene unresolved -> empty index -> every read []
    ix = Indexer.from_gene_slices([None, None, None, None])
    ix.make_index()
    assert ix.info()["n_keys"] == 0 and ix.m_fusion_seq == ["", "", "", ""]
    r1 = (b"CATCACACACCTTGACTGGTCCCCAGACAACAAGTATATAATGTCTAACTCGGGAGACTATGAAATATTGTACTGTAAGTATGAATGATT"
          b"TTATATATATATATATATGCTATGATTATATTTATATATATAATAATTATTTTCCATATAT")
    assert ix.map_reads([r1, rc(r1), r1[:148]]) == [[], [], []]
    ix.close()
    # no genes at all
    ix = Indexer.from_gene_slices([])
    ix.make_index()
    assert ix.map_reads([r1]) == [[]]
    ix.close()


def test_with_loaded_ref_constructor(gpu_device, oracle):
    """Indexer::with_loaded_ref + make_index chromosome resolution and slicing
    (indexer.rs:137-159): exact name, "chr"+name, name without "chr", missing."""
    from genefuserust_amd import Fusion, Gene, Indexer
    rng = np.random.default_rng(21)
    ref = {"chr2": rand_seq(rng, 4000), "7": rand_seq(rng, 3000), "chrX": rand_seq(rng, 2500).lower()}
    fus = [Fusion(Gene("A", "chr2", 100, 1900)), Fusion(Gene("B", "chr7", 50, 1500, True)),
           Fusion(Gene("C", "X", 10, 2000)), Fusion(Gene("D", "chr9", 1, 1000))]
    ix = Indexer.with_loaded_ref(ref, fus)
    ix.make_index()
    slices = [ref["chr2"][100:1900], ref["7"][50:1500], ref["chrX"][10:2000], None]
    ox = oracle.OracleIndexer(slices)
    assert ix.m_fusion_seq == [ox.fusion_seq(c) for c in range(4)]
    assert ix.m_fusion_seq[2] == ref["chrX"][10:2000].decode().upper() and ix.m_fusion_seq[3] == ""
    up = [s.upper() if s else s for s in slices]
    reads = [up[0][500:575] + up[1][700:775], up[2][300:380] + up[0][1000:1070], rc(up[1][200:275] + up[2][900:975])]
    _compare_reads(ix, ox, reads, "with_loaded_ref")
    ix.close()


@pytest.mark.parametrize("variant", [0, 1, 2, "0-nofilter", "0-midfilter", "0-bigfilter"])
@pytest.mark.parametrize("shape,scale,n_reads", [("IDX-T", 0.02, 60000), ("IDX-C", 0.004, 60000)])
def test_synthetic_parity_medium(gpu_device, oracle, shape, scale, n_reads, variant, monkeypatch):
    """Repeat-rich synthetic genes (2 % repeat family, N bases) and a junction-heavy
    read mix: every read's SeqMatch list equals the oracle's."""
    from genefuserust_amd import Indexer
    from genefuserust_amd import synth
    genes = synth.make_geneset(shape, scale=scale)
    if variant == "0-nofilter":
        monkeypatch.setenv("GF_BLOOM_KIB", "0")
        variant = 0
    elif variant == "0-midfilter":  # indexes whose filter outgrows the L2 but is still used inline
        monkeypatch.setenv("GF_BLOOM_KIB", "1")
        variant = 0
    elif variant == "0-bigfilter":  # larger still: Infinity-Cache filter, used by the filter kernel only
        monkeypatch.setenv("GF_BLOOM_KIB", "1")
        monkeypatch.setenv("GF_BLOOM_MID_KIB", "1")
        variant = 0
    ix = Indexer.from_gene_slices(genes.seqs, genes.reversed_flags)
    ix.make_index()
    ix.set_map_variant(variant)
    ox = oracle.OracleIndexer(genes.seqs)
    info = ix.info()
    st = ox.stats()
    assert (info["n_keys"], info["n_high_keys"], info["n_unique"]) == (st["n_keys"], st["n_high_keys"], st["m_unique_pos"])
    keys = ox.keys()[::97]
    cnt, ctg, pos = ix.lookup(keys.astype(np.uint32))
    for j, k in enumerate(keys):
        n, sites = ox.lookup(int(k))
        assert cnt[j] == n
        if n > 0:
            assert [(int(ctg[j, t]), int(pos[j, t])) for t in range(n)] == sites
    synth.MIXES["TEST"] = (0.2, 0.5, 0.3)
    total_hits = 0
    for L in (150, 100, 251, 272, 320):  # 272 = two merged 151-base reads; 320 = the flat kernels' limit
        rb = synth.make_reads(genes, n_reads // 3, read_len=L, mix="TEST", seed=77 + L)
        bases = rb.bases.numpy()
        offsets = rb.offsets.numpy()
        counts, matches = ix.map_reads_packed(bases, offsets)
        ocounts, omatches = ox.map_reads_packed(bases, offsets, threads=8)
        assert (counts == ocounts).all(), np.nonzero(counts != ocounts)[0][:10]
        nz = counts > 0
        assert (matches[nz, 0] == omatches[nz, 0]).all()
        two = counts == 2
        assert (matches[two, 1] == omatches[two, 1]).all()
        total_hits += int(nz.sum())
    assert total_hits > 1000  # the mix really exercises the second pass
    ix.close()


@pytest.mark.parametrize("shape,scale", [("IDX-T", 0.3), ("IDX-C", 0.03)])
def test_repeat_rich_genes_parity(gpu_device, oracle, shape, scale):
    """Genes of which 30 % are copies of a 20-element repeat family and 5 % poly-A / tandem repeats: reads inside a
    repeat show nothing but keys with six sites or more (HIGH: they cannot vote, indexer.rs:202-239), reads across a
    repeat's edge mix them with unique keys.  The bucket pass proves whole stretches of such windows unable to vote from
    ONE probe (the key's representative site + the per-site HIGH flags, r04): every read's result equals the oracle's."""
    from genefuserust_amd import Indexer, synth
    genes = synth.make_geneset(shape, scale=scale, repeat_frac=0.3, low_complexity_frac=0.05)
    ix = Indexer.from_gene_slices(genes.seqs, genes.reversed_flags)
    ix.make_index()
    ox = oracle.OracleIndexer(genes.seqs)
    info, st = ix.info(), ox.stats()
    assert (info["n_keys"], info["n_high_keys"], info["n_unique"]) == (st["n_keys"], st["n_high_keys"], st["m_unique_pos"])
    assert info["n_high_keys"] > 200
    synth.MIXES["TEST"] = (0.1, 0.6, 0.3)
    total_hits = 0
    for L in (150, 100, 251):
        rb = synth.make_reads(genes, 40000, read_len=L, mix="TEST", seed=31 + L)
        bases, offsets = rb.bases.numpy(), rb.offsets.numpy()
        counts, matches = ix.map_reads_packed(bases, offsets)
        ocounts, omatches = ox.map_reads_packed(bases, offsets, threads=8)
        assert (counts == ocounts).all(), np.nonzero(counts != ocounts)[0][:10]
        nz = counts > 0
        assert (matches[nz, 0] == omatches[nz, 0]).all()
        two = counts == 2
        assert (matches[two, 1] == omatches[two, 1]).all()
        total_hits += int(nz.sum())
    assert total_hits > 500
    ix.close()


def test_cpp_host_mirror(gpu_device, tmp_path):
    """include/gf_indexer.hpp (the compiled-language host side above the C ABI):
    planted-fusion known answer through the C++ Indexer mirror."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "test_indexer")
    subprocess.run(["g++", "-std=c++17", "-O1", "-I" + os.path.join(root, "include"),
                    os.path.join(root, "tests", "cpp", "test_indexer.cpp"), "-o", exe,
                    "-L" + os.path.join(root, "genefuserust_amd"), "-lgfmatch",
                    "-Wl,-rpath," + os.path.join(root, "genefuserust_amd")], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout + out.stderr


def test_full_size_properties(gpu_device, oracle):
    """BASELINE configs[1] at full size (20 M reads of 150 bp vs the druggable-shaped
    index), checked through size-independent properties:
      * every read the GPU reports segments for is re-mapped by the oracle (identical),
        and so is a 100 K sample of the reads it reports nothing for;
      * order independence: the same reads in reversed order give the reversed result
        (no cross-read interference), compared as a checksum of per-read checksums;
      * compaction: hit list = exactly the non-zero counts, ascending;
      * a second launch is bit-identical (determinism)."""
    import torch
    from genefuserust_amd import Indexer, synth
    from genefuserust_amd.indexer import hits_to_numpy
    n, L = 20_000_000, 150
    genes = synth.make_geneset("IDX-D")
    ix = Indexer.from_gene_slices(genes.seqs, genes.reversed_flags)
    ix.make_index()
    ox = oracle.OracleIndexer(genes.seqs)
    st, info = ox.stats(), ix.info()
    assert (info["n_keys"], info["n_high_keys"], info["n_unique"]) == (st["n_keys"], st["n_high_keys"], st["m_unique_pos"])
    rb = synth.make_reads(genes, n, read_len=L, mix="PANEL", seed=4242, device="cuda")
    counts, matches = ix.map_reads_device(rb.bases, rb.offsets, L)
    torch.cuda.synchronize()
    assert int((counts > 2).sum()) == 0
    hit_idx = torch.nonzero(counts).flatten()
    assert 5_000 < hit_idx.numel() < 200_000
    # oracle on all hits + a sample of non-hits
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    sample = torch.randint(0, n, (100_000,), device="cuda", generator=gen)
    sel = torch.unique(torch.cat([hit_idx, sample]))
    reads2d = rb.bases.view(n, L)
    sub = reads2d[sel].contiguous().cpu().numpy().reshape(-1)
    offs = np.arange(sel.numel() + 1, dtype=np.int64) * L
    oc, om = ox.map_reads_packed(sub, offs, threads=16)
    gc = counts[sel].cpu().numpy().astype(np.int32)
    gm = matches[sel].cpu().numpy().view(om.dtype).reshape(-1, 2)
    assert (gc == oc).all(), "count mismatch at %s" % sel.cpu().numpy()[np.nonzero(gc != oc)[0][:5]]
    assert (gm[oc > 0, 0] == om[oc > 0, 0]).all() and (gm[oc == 2, 1] == om[oc == 2, 1]).all()
    assert int((oc == 2).sum()) > 1000
    # checksum of per-read checksums, order independence, determinism
    def digest(c, m):
        mm = m.view(-1, 8).to(torch.int64)
        valid1 = (c >= 1).to(torch.int64)[:, None]
        valid2 = (c == 2).to(torch.int64)[:, None]
        w = torch.tensor([3, 5, 7, 11], dtype=torch.int64, device=c.device)
        per = c.to(torch.int64) * 1000003 + ((mm[:, :4] * w) * valid1).sum(1) + ((mm[:, 4:] * w * 13) * valid2).sum(1)
        return per
    d1 = digest(counts, matches)
    c2, m2 = ix.map_reads_device(rb.bases, rb.offsets, L)
    torch.cuda.synchronize()
    assert torch.equal(digest(c2, m2), d1)
    rev_bases = reads2d.flip(0).contiguous().view(-1)
    c3, m3 = ix.map_reads_device(rev_bases, rb.offsets, L)
    torch.cuda.synchronize()
    assert torch.equal(digest(c3, m3).flip(0), d1)
    # compaction
    hits, n_hits = ix.compact_hits_device(counts, matches, n, read_id_base=7, cap=hit_idx.numel() + 10)
    torch.cuda.synchronize()
    h = hits_to_numpy(hits[: int(n_hits.item())])
    assert int(n_hits.item()) == hit_idx.numel()
    assert (h["read_id"] == hit_idx.cpu().numpy() + 7).all()
    assert (h["n"] == counts[hit_idx].cpu().numpy()).all()
    ix.close()


@pytest.mark.parametrize("variant", [0, "0-pack", 2])
def test_batches_with_gaps_and_mixed_lengths(branch_index, golden, oracle, variant):
    """Reads need not be packed back to back: gaps between reads (so that most of the
    batch lies beyond the packed stream of the flat pipeline), a first offset > 0, every
    length class in one batch, junk bytes in the gaps."""
    import torch
    from tests.helpers import matches_to_tuples
    reads = [c["read"].encode() for c in golden["cases"]]
    want = [[tuple(m) for m in c["expect"]] for c in golden["cases"]]
    rng = np.random.default_rng(3)
    long1 = reads[0] * 3          # 450 bases -> 1024 class
    long2 = (reads[3] + reads[4]) * 4  # 1200 bases -> 4096 class
    ox = oracle.OracleIndexer([None if x is None else x.encode() for x in golden["genes"]])
    batch = reads + [long1, long2]
    want = want + [ox.map_read(long1), ox.map_read(long2)]
    branch_index.set_pack_call_reads(-1 if variant == "0-pack" else 0)  # "0-pack": the zero-copy route of a host call
    branch_index.set_map_variant(0 if variant == "0-pack" else variant)
    for gap in (0, 7, 400):
        buf = bytearray(rng.integers(65, 91, size=37, dtype=np.uint8).tobytes())  # junk before the first read
        offs = []
        for r in batch:
            offs.append(len(buf))
            buf += r
            # a read's bytes end where the next begins in `offsets`, so the junk belongs to the
            # gap only if the offsets array skips it: emulate with zero-length "reads" = gaps
            if gap:
                offs.append(len(buf))
                buf += rng.integers(65, 91, size=gap, dtype=np.uint8).tobytes()
        offs.append(len(buf))
        bases = np.frombuffer(bytes(buf), dtype=np.uint8)
        offsets = np.array(offs, dtype=np.int64)
        counts, matches = branch_index.map_reads_packed(bases, offsets)
        got = matches_to_tuples(counts, matches)
        if gap:
            # odd entries are the junk "reads": random upper-case letters, map to []
            assert all(g == [] for g in got[1::2]) or gap >= 54  # junk longer than 53 is mapped like any read
            got = got[0::2]
        assert got == want, (variant, gap)
    branch_index.set_map_variant(0)
    branch_index.set_pack_call_reads(-1)


def test_device_offsets_with_holes(branch_index, golden):
    """Device API with an offsets array whose reads are far apart (holes the offsets skip are
    not expressible in one array, so use long junk reads as holes and a tight max_read_len):
    reads beyond the packed stream of the flat pipeline are routed to the exact kernel."""
    import torch
    from tests.helpers import matches_to_tuples
    reads = [c["read"].encode() for c in golden["cases"][:60]]
    want = [[tuple(m) for m in c["expect"]] for c in golden["cases"][:60]]
    junk = b"N" * 3000
    buf, offs = bytearray(), []
    for r in reads:
        offs.append(len(buf)); buf += r
        offs.append(len(buf)); buf += junk
    offs.append(len(buf))
    d_b = torch.from_numpy(np.frombuffer(bytes(buf), dtype=np.uint8).copy()).cuda()
    d_o = torch.from_numpy(np.array(offs, dtype=np.int64)).cuda()
    # the batch limit covers the real reads only: junk entries are marked too long, and the
    # packed stream (n * 256 bases) ends long before most real reads start
    counts, matches = branch_index.map_reads_device(d_b, d_o, 256)
    torch.cuda.synchronize()
    c = counts.cpu().numpy().astype(np.int32)
    assert (c[1::2] == 255).all()
    m = matches.cpu().numpy().view(np.dtype([("seq_start", "<i4"), ("seq_end", "<i4"), ("position", "<i4"),
                                            ("contig", "<i2"), ("pad", "<i2")])).reshape(-1, 2)
    lens = np.array([len(r) for r in reads])
    cr = c[0::2].copy()
    assert (cr[lens > 256] == 255).all()
    cr[lens > 256] = 0
    got = matches_to_tuples(cr, m[0::2])
    for k, (g, w, ln) in enumerate(zip(got, want, lens)):
        if ln <= 256:
            assert g == w, k


@pytest.mark.parametrize("max_read_len", [160, 256])
def test_read_larger_than_a_staging_tile(branch_index, golden, max_read_len):
    """The fused seed+verify kernel stages the next reads' bytes in LDS tile by tile; a read
    larger than a whole tile (40 KB / 64 KB) must be stepped over without staging, and the
    reads around it keep their answers."""
    import torch
    from tests.helpers import matches_to_tuples
    cases = [c for c in golden["cases"] if len(c["read"]) <= max_read_len]
    reads = [c["read"].encode() for c in cases]
    want = [[tuple(m) for m in c["expect"]] for c in cases]
    giant = (b"ACGTTGCA" * 9000)[:70001]
    batch = reads[:70] + [giant] + reads[70:] + [giant, giant] + reads[:5]
    want = want[:70] + [None] + want[70:] + [None, None] + want[:5]
    offs = np.zeros(len(batch) + 1, dtype=np.int64)
    np.cumsum([len(r) for r in batch], out=offs[1:])
    d_b = torch.from_numpy(np.frombuffer(b"".join(batch), dtype=np.uint8).copy()).cuda()
    d_o = torch.from_numpy(offs).cuda()
    counts, matches = branch_index.map_reads_device(d_b, d_o, max_read_len)
    torch.cuda.synchronize()
    c = counts.cpu().numpy().astype(np.int32)
    m = matches.cpu().numpy().view(np.dtype([("seq_start", "<i4"), ("seq_end", "<i4"), ("position", "<i4"),
                                            ("contig", "<i2"), ("pad", "<i2")])).reshape(-1, 2)
    for k, w in enumerate(want):
        if w is None:
            assert c[k] == 255, k
        else:
            assert matches_to_tuples(c[k:k + 1], m[k:k + 1]) == [w], k


def test_single_read_calls_are_stable(branch_index, golden):
    """Indexer::map_read is called one read at a time by the reference; 400 back-to-back
    n = 1 calls (tiny kernels, workspace reused immediately) must keep returning the golden
    answers.  Regression: a stream-ordered free used to race with the kernels here."""
    cases = [c for c in golden["cases"] if c["expect"]][:20]
    for rep in range(20):
        for c in cases:
            got = branch_index.map_read(c["read"])
            flat = [(m.seq_start, m.seq_end, m.start_gp.contig, m.start_gp.position) for m in got]
            assert flat == [tuple(m) for m in c["expect"]], (rep, c["label"])


def test_segment_mask_device_vs_reference_scan_on_random_masks(branch_index, oracle):
    """The device form of segment_mask (heads / targets as bit masks, gf_map_kernels.h) against the
    reference's sequential scan (indexer.rs:616-679, oracle: orc_segment_mask) on masks that no read
    has to produce: random class runs with every gap length around ALLOWED_GAP = 10, higher classes
    cutting runs, runs starting at the last base, ties between equally long runs, lengths 1..4096."""
    from genefuserust_amd import _lib
    rng = np.random.default_rng(2024)
    masks, want = [], []
    lens = [1, 2, 21, 22, 23, 63, 64, 65, 127, 128, 129, 150, 255, 256, 257, 272, 320, 1023, 1024, 1025, 4095, 4096]
    lens += [int(x) for x in rng.integers(20, 400, size=1500)] + [int(x) for x in rng.integers(400, 4097, size=120)]
    for L in lens:
        style = int(rng.integers(0, 5))
        m = np.zeros(L, dtype=np.uint8)
        if style == 0:     # i.i.d. classes
            m = rng.integers(0, 4, size=L).astype(np.uint8)
        elif style == 4:   # one class everywhere, or nothing
            m[:] = int(rng.integers(0, 4))
        else:              # runs of a class separated by gaps of 0..14 low-class positions
            p = 0
            while p < L:
                run = int(rng.integers(1, 60 if style < 3 else 12))
                cls = int(rng.choice([2, 3, 3, 2, 1])) if style != 2 else int(rng.choice([2, 3]))
                m[p:p + run] = cls
                p += run
                gap = int(rng.integers(0, 15))
                g = min(gap, L - p)
                if g > 0:
                    m[p:p + g] = rng.integers(0, 2, size=g)
                p += gap
            if rng.random() < 0.3:
                m[L - 1] = int(rng.choice([2, 3]))   # a run beginning at the last base is never seen (:635-640)
        masks.append(m)
    gp1 = np.array([oracle.gp_to_i64(int(c), int(p)) for c, p in zip(rng.integers(0, 30, len(masks)), rng.integers(-5000, 5000, len(masks)))], dtype=np.int64)
    gp2 = np.array([oracle.gp_to_i64(int(c), int(p)) for c, p in zip(rng.integers(0, 30, len(masks)), rng.integers(-5000, 5000, len(masks)))], dtype=np.int64)
    for m, a, b in zip(masks, gp1, gp2):
        want.append(oracle.segment_mask(m, oracle.i64_to_gp(int(a)), oracle.i64_to_gp(int(b))))
    flat = np.concatenate(masks)
    offs = np.zeros(len(masks) + 1, dtype=np.int64)
    np.cumsum([m.size for m in masks], out=offs[1:])
    counts = np.zeros(len(masks), dtype=np.int32)
    out = np.zeros((len(masks), 2), dtype=_lib.SEQMATCH_DTYPE)
    _lib.check(_lib.lib().gf_segment_mask_test(branch_index._handle(), flat.ctypes.data, offs.ctypes.data, len(masks),
                                               gp1.ctypes.data, gp2.ctypes.data, counts.ctypes.data, out.ctypes.data))
    got = matches_to_tuples(counts, out)
    bad = [k for k in range(len(masks)) if got[k] != want[k]]
    assert not bad, (len(bad), bad[0], got[bad[0]], want[bad[0]], masks[bad[0]].tolist())
    assert sum(1 for w in want if len(w) == 2) > 300 and sum(1 for w in want if len(w) == 1) > 100


def _dense_equal(a, b, n):
    ca, ma = a
    cb, mb = b
    import torch
    assert torch.equal(ca[:n], cb[:n])
    k = ca[:n].to(torch.int64)
    k = torch.where(k <= 2, k, torch.zeros_like(k))   # (255 = longer than the batch limit: nothing written)
    valid = (torch.arange(2, device=ca.device)[None, :] < k[:, None])[:, :, None]
    assert torch.equal(ma[:n] * valid, mb[:n] * valid)


def test_packed_hand_over_equals_ascii(branch_index, golden, gpu_device):
    """gf_pack_bases_device + gf_map_reads_packed_device return what gf_map_reads_device returns on the
    ASCII buffer: golden branch cases with every length class (flat kernels, 1024- and 4096-base lists),
    every phase of the first read inside its 16-base chunk, N / lower-case bases, empty reads."""
    import torch
    from genefuserust_amd.synth import ragged_batch
    reads = [c["read"].encode() for c in golden["cases"]]
    reads += [reads[0] * 3, (reads[3] + reads[4]) * 4, b"", b"acgtnACGTN" * 20, b"N" * 70]
    for lead in (0, 1, 5, 15, 16, 23):
        bases, offsets = ragged_batch(reads)
        bases = np.concatenate([np.frombuffer(b"T" * lead, dtype=np.uint8), bases])
        offsets = offsets + lead
        d_b, d_o = torch.from_numpy(bases).cuda(), torch.from_numpy(offsets).cuda()
        pk, iv = branch_index.pack_bases_device(d_b)
        for lcap in (270, 1024, 4096):
            a = branch_index.map_reads_device(d_b, d_o, lcap)
            b = branch_index.map_reads_packed_device(pk, iv, d_o, lcap)
            torch.cuda.synchronize()
            _dense_equal(a, b, len(reads))
    # the packed form itself, against a plain restatement: base j of chunk c in bits 2j, 2j+1 (A0 C1 T2 G3)
    code = {65: 0, 67: 1, 84: 2, 71: 3}
    host = bases
    pkh, ivh = pk.cpu().numpy().view(np.uint32), iv.cpu().numpy().view(np.uint16)
    for c in (0, 1, 7, len(host) // 16 - 1, len(host) // 16):
        w = bad = 0
        for j in range(16):
            k = 16 * c + j
            if k < len(host) and int(host[k]) in code:
                w |= code[int(host[k])] << (2 * j)
            else:
                bad |= 1 << j
        assert int(ivh[c]) == bad and (int(pkh[c]) & ~sum(3 << (2 * j) for j in range(16) if bad >> j & 1)) == w, c


def test_packed_hand_over_synthetic_and_rate(gpu_device):
    """2 M synthetic reads of 150 and 251 bases: packed and ASCII paths bit-identical; prints both rates."""
    import torch
    from genefuserust_amd import Indexer, synth
    genes = synth.make_geneset("IDX-D", scale=0.25)
    ix = Indexer.from_gene_slices(genes.seqs, genes.reversed_flags)
    ix.make_index()
    for L in (150, 251):
        n = 2_000_000
        rb = synth.make_reads(genes, n, read_len=L, mix="PANEL", seed=99 + L, device="cuda")
        pk, iv = ix.pack_bases_device(rb.bases)
        a = ix.map_reads_device(rb.bases, rb.offsets, L)
        b = ix.map_reads_packed_device(pk, iv, rb.offsets, L)
        torch.cuda.synchronize()
        _dense_equal(a, b, n)
        assert int((a[0] > 0).sum()) > 500
    ix.close()


@pytest.mark.gpu
@pytest.mark.parametrize("L", [64, 100, 150, 151, 160, 161, 250, 320])
def test_fixed_length_batches_equal_the_offsets_form(gpu_device, L):
    """gf_map_reads_fixed_device (offsets computed, not loaded) == gf_map_reads_device with offsets[r] = r * L, for every
    word class of the flat pipeline, batch sizes that are no multiple of a wavefront, N and lower case in the reads."""
    import torch
    from genefuserust_amd import Indexer, synth
    genes = synth.make_geneset("IDX-T", scale=0.05)
    ix = Indexer.from_gene_slices(genes.seqs, genes.reversed_flags)
    ix.make_index()
    synth.MIXES["TEST"] = (0.3, 0.4, 0.3)
    n = 70_001
    rb = synth.make_reads(genes, n, read_len=L, mix="TEST", seed=100 + L, device="cuda")
    bases = rb.bases.clone()
    bases[torch.randint(0, bases.numel(), (200,), device="cuda")] = ord("n")
    c0, m0 = ix.map_reads_device(bases, rb.offsets, L)
    c1, m1 = ix.map_reads_fixed_device(bases, L)
    assert L < 100 or int((c0 > 0).sum()) > 1000   # (a 54-base read has 20 windows: nothing passes the gate)
    assert torch.equal(c0, c1)
    nz = (c0 > 0).nonzero().flatten()
    assert torch.equal(m0[nz, 0], m1[nz, 0])
    two = (c0 == 2).nonzero().flatten()
    assert torch.equal(m0[two, 1], m1[two, 1])
    ix.close()
