"""The inputs of Indexer::make_index — Fusion::parse_csv / Gene (fusion.rs:22-86, gene.rs) and
FastaReader::read_all (fasta_reader.rs:117-200) — in the host-side mirror, pinned by the
reference's own tests: the four pos2str strings of fusion.rs:112-150 on testdata/fusions.csv
and the two contigs of fasta_reader.rs:233-258 on testdata/tinyref.fa(.gz) (the files are
kept as data under tests/golden/)."""
import os

import pytest

from oracle import indexer_model as M

HERE = os.path.dirname(__file__)
CSV = os.path.join(HERE, "golden", "fusions.csv")
FA = os.path.join(HERE, "golden", "tinyref.fa")
FAGZ = os.path.join(HERE, "golden", "tinyref.fa.gz")

CONTIG1 = b"GATCACAGGTCTATCACCCTATTAATTGGTATTTTCGTCTGGGGGGTGTGGAGCCGGAGCACCCTATGTCGCAGT"
CONTIG2 = b"GTCTGCACAGCCGCTTTCCACACAGAACCCCCCCCTCCCCCCGCTTCTGGCAAACCCCAAAAACAAAGAACCCTA"


def test_parse_csv_reference_vectors():
    from genefuserust_amd.indexer import Fusion
    fusions = Fusion.parse_csv(CSV)
    by_name = {f.m_gene.m_name: f for f in fusions}
    assert [f.m_gene.m_name for f in fusions] == ["ALK", "ROS1", "RET", "EML4"]
    # fusion.rs:121-141
    assert by_name["ALK"].pos2str(-30582) == "ALK:exon:20|-chr2:29446222"
    assert by_name["ALK"].pos2str(31060) == "ALK:intron:19|+chr2:29446700"
    assert by_name["EML4"].pos2str(95365) == "EML4:exon:6|+chr2:42491855"
    assert by_name["EML4"].pos2str(95346) == "EML4:intron:5|+chr2:42491836"
    assert by_name["ALK"].is_reversed() and not by_name["EML4"].is_reversed()
    # the independent model agrees on every gene and on a sweep of positions
    text = open(CSV).read()
    model = M.csv_genes(text)
    assert [(f.m_gene.m_name, f.m_gene.m_chr, f.m_gene.m_start, f.m_gene.m_end, f.m_gene.m_reversed,
             [(e.id, e.start, e.end) for e in f.m_gene.m_exons]) for f in fusions] == \
        [(g[0], g[1], g[2], g[3], g[5], g[4]) for g in model]
    for f, g in zip(fusions, model):
        span = f.m_gene.m_end - f.m_gene.m_start
        for pos in list(range(-span, span, max(1, span // 400))) + [0, 1, -1]:
            assert f.pos2str(pos) == M.gene_pos2str(g, pos)
            assert f.m_gene.gene_pos_2_chr_pos(pos) == (abs(pos) + f.m_gene.m_start) * (-1 if pos < 0 else 1)


def test_parse_csv_edges():
    from genefuserust_amd.indexer import Fusion
    text = "#comment,line\n\n>G1, chr9:100-900\n1,100,200\r\n2,300,400\nnot-a-line\n>BAD,nocolon\n5,1,2\n" \
           ">G2,chrX:5000-1000, extra\n3,900,950\n2,700,800\n1,,\n"
    with pytest.raises(ValueError):
        Fusion.parse_csv_text(text)           # "1,," : an exon line whose numbers do not parse is an error
    fusions = Fusion.parse_csv_text(text.replace("1,,\n", ""))
    assert [(f.m_gene.m_name, f.m_gene.m_chr, f.m_gene.m_start, f.m_gene.m_end, f.m_gene.m_reversed) for f in fusions] == \
        [("G1", "chr9", 100, 900, False), ("G2", "chrX", 5000, 1000, True)]
    # the exons after the invalid ">BAD" line belong to the invalid gene, which is dropped
    assert [(e.id, e.start, e.end) for e in fusions[0].m_gene.m_exons] == [(1, 100, 200), (2, 300, 400)]
    assert [(g[0], g[5]) for g in M.csv_genes(text.replace("1,,\n", ""))] == [("G1", False), ("G2", True)]
    assert fusions[1].m_gene.get_exon_intron(-(950 - 5000)) in ((None, None), (True, 3))


def test_fasta_reader_reference_vectors():
    from genefuserust_amd.indexer import FastaReader
    for path in (FAGZ, FA):
        r = FastaReader(path, True)
        r.read_all()
        assert r.m_all_contigs["contig1"] == CONTIG1 and r.m_all_contigs["contig2"] == CONTIG2  # fasta_reader.rs:237-238
        assert r.m_all_contigs == M.fasta_contigs(r._data, True)


def test_fasta_reader_edges(tmp_path):
    from genefuserust_amd.indexer import FastaReader
    cases = [b">a\nACGT\nacgt\n>b desc ription\nNN-*\n12\n>c\n", b"junk>x\nAC>y\nGT>", b">", b">only", b">>a\nA\n"]
    for k, data in enumerate(cases):
        p = tmp_path / ("t%d.fa" % k)
        p.write_bytes(data)
        for upper in (True, False):
            r = FastaReader(str(p), upper)
            r.read_all()
            assert r.m_all_contigs == M.fasta_contigs(data, upper), (data, upper)
    r = FastaReader(str(tmp_path / "t0.fa"), True)
    r.read_all()
    # the description after a blank is not part of the name — and, as in the reference, its
    # letters end up in front of the sequence
    assert r.m_all_contigs == {"a": b"ACGTACGT", "b": b"DESCRIPTIONNN-*", "c": b""}
    (tmp_path / "e.fa").write_bytes(b"")
    with pytest.raises(ValueError):
        FastaReader(str(tmp_path / "e.fa"), True)


def test_config0_gene_slices_resolve_to_nothing():
    """BASELINE configs[0]: testdata/fusions.csv names chr2/chr10 genes, testdata/tinyref.fa has
    contig1/contig2 only: every gene is unresolved (indexer.rs:137-150) and indexes nothing."""
    from genefuserust_amd.indexer import FastaReader, Fusion, resolve_gene_slice
    r = FastaReader(FA, True)
    r.read_all()
    fusions = Fusion.parse_csv(CSV)
    assert len(fusions) >= 2 and all(resolve_gene_slice(r.m_all_contigs, f.m_gene) is None for f in fusions)


@pytest.mark.gpu
def test_config0_files_end_to_end(gpu_device):
    """BASELINE configs[0] on the device: testdata R1.fq / R2.fq + tinyref.fa + fusions.csv.
    No gene resolves, the index is empty (zero keys), every pair maps to nothing — and the
    same plumbing with a reference that does hold the genes gives the answers of an index
    built from the slices directly."""
    import numpy as np
    import torch
    from genefuserust_amd import Indexer
    from genefuserust_amd.fastq import FastqReaderPair
    from genefuserust_amd.indexer import FastaReader, Fusion, Gene
    from genefuserust_amd.read_pair import fast_merge_device
    from tests.helpers import rand_seq
    g = os.path.join(HERE, "golden")
    ref = FastaReader(FA, True)
    ref.read_all()
    fusions = Fusion.parse_csv(CSV)
    ix = Indexer(ref.m_all_contigs, fusions)
    ix.make_index()
    assert ix.info()["n_keys"] == 0 and ix.m_fusion_seq == [""] * len(fusions)
    (l, _), (r, _) = FastqReaderPair.from_paths(os.path.join(g, "R1.fq"), os.path.join(g, "R2.fq")).read_all_device(ix)
    assert l.n_records == r.n_records == 3
    for b in (l, r):
        counts, _ = ix.map_reads_device(b.bases, b.offsets, b.max_read_len())
        assert int(counts.sum()) == 0
    bases, quals, off, diff = fast_merge_device(ix, l.bases, l.quals, l.offsets, r.bases, r.quals, r.offsets, 151)
    counts, _ = ix.map_reads_device(bases, off, 320)
    torch.cuda.synchronize()
    assert int(counts.sum()) == 0
    ix.close()
    # a reference that holds the genes: "chr"-prefix fallbacks of indexer.rs:137-152 included
    rng = np.random.default_rng(8)
    contigs = {"chr7": rand_seq(rng, 6000), "12": rand_seq(rng, 5000)}
    genes = [Fusion(Gene("GA", "7", 500, 3500)), Fusion(Gene("GB", "chr12", 1000, 4000)), Fusion(Gene("GC", "chr3", 1, 2))]
    a = Indexer(contigs, genes)
    a.make_index()
    b = Indexer.from_gene_slices([contigs["chr7"][500:3500], contigs["12"][1000:4000], None])
    b.make_index()
    assert a.info()["n_keys"] == b.info()["n_keys"] > 5000 and a.m_fusion_seq[2] == ""
    read = contigs["chr7"][1000:1075] + contigs["12"][2000:2075]
    assert a.map_read(read) == b.map_read(read) and len(a.map_read(read)) == 2
    a.close()
    b.close()


def test_cpp_mirror_of_the_inputs(tmp_path):
    """include/gf_indexer.hpp: the same parsers in the C++ mirror, same reference vectors (host only)."""
    import subprocess
    root = os.path.dirname(HERE)
    exe = str(tmp_path / "test_inputs")
    subprocess.run(["g++", "-O1", "-std=c++17", "-I", os.path.join(root, "include"),
                    os.path.join(HERE, "cpp", "test_inputs.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe, CSV, FA], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.gpu
def test_scan_pair_end_files(gpu_device, tmp_path):
    """Files in, sorted ReadMatch list out: a FASTA with two genes, a fusion CSV, and paired FASTQ
    files cut from planted fusions (plus background).  The matches must name the planted pairs,
    sit at the planted breakpoints, and come out in the sort_matches order."""
    import gzip
    import numpy as np
    from genefuserust_amd.scan import scan_pair_end_files
    from tests.helpers import rand_seq, rc
    rng = np.random.default_rng(12)
    chr1, chr2 = rand_seq(rng, 9000), rand_seq(rng, 8000)
    fa = tmp_path / "ref.fa"
    fa.write_bytes(b">chr1 test\n" + b"\n".join(chr1[i:i + 60] for i in range(0, len(chr1), 60)) +
                   b"\n>chr2\n" + chr2.lower() + b"\n")
    csv = tmp_path / "f.csv"
    csv.write_text(">GA,chr1:1000-7000\n1,1000,3000\n2,4000,7000\n\n>GB,chr2:500-6500\n1,500,2500\n2,3500,6500\n")
    # "chr1 test": the description's letters are prepended to the sequence by the reference's reader
    shift = len(b"TEST")
    ga = (b"TEST" + chr1)[1000:7000]
    gb = chr2.upper()[500:6500]
    l_txt, r_txt, planted = [], [], {}
    for k in range(60):
        if k % 3 == 0:
            p, q = int(rng.integers(400, 5500)), int(rng.integers(400, 5500))
            frag = ga[p - 140:p] + gb[q:q + 140]
            planted[b"@pair%d/1" % k] = (p - 1, q)
        else:
            frag = rand_seq(rng, 280)
        flen = int(rng.integers(200, 281))
        lo = (280 - flen) // 2
        f = frag[lo:lo + flen]
        s1, s2 = f[:150], rc(f)[:150]
        l_txt += [b"@pair%d/1" % k, s1, b"+", b"F" * len(s1)]
        r_txt += [b"@pair%d/2" % k, s2, b"+", b"F" * len(s2)]
    r1, r2 = tmp_path / "R1.fq", tmp_path / "R2.fq.gz"
    r1.write_bytes(b"\n".join(l_txt) + b"\n")
    with gzip.open(r2, "wb") as f:
        f.write(b"\n".join(r_txt) + b"\n")
    kept, counters = scan_pair_end_files(str(fa), str(csv), str(r1), str(r2))
    assert counters["pairs"] == 60 and shift == 4
    names = {m.m_name.split(b" merged_diff_")[0] for m in kept}  # (merged reads carry read.rs:372's suffix)
    assert names <= set(planted) and len(names) >= 15
    assert any(b" merged_diff_" in m.m_name for m in kept)
    for m in kept:
        p_last, q_first = planted[m.m_name.split(b" merged_diff_")[0]]
        assert len(m.m_quality) == len(m.m_read)
        left, right = (m.m_left_gp, m.m_right_gp)
        assert {left.contig, right.contig} == {0, 1}
        a = left if left.contig == 0 else right
        b = right if left.contig == 0 else left
        # the breakpoint may slide by a few bases where the flanks agree by chance
        assert abs(abs(a.position) - p_last) <= 4 and abs(abs(b.position) - q_first) <= 4
    order = [(m.m_read_break, len(m.m_read), m.m_name) for m in kept]
    assert order == sorted(order, key=lambda t: (-t[0], t[1], tuple(-c for c in t[2])))
    # with the reference's last filter step as it is (matcher.py): on a reference this small its
    # Matcher finds a key with a handful of sites, votes, and panics on the first window whose
    # key it does not hold — the reference binary would crash on these files
    from genefuserust_amd import MatcherPanic
    with pytest.raises(MatcherPanic):
        scan_pair_end_files(str(fa), str(csv), str(r1), str(r2), remove_alignables=True)
