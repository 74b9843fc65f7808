"""The index's derived device arrays, word by word: the genes of both strands in site-code space, the
"only site of its key" / "one of six or more" flags and the presence filter over canonical 14-mers (csrc/gf_index_kernels.h,
layout in csrc/gf_table.h) are what the mapping kernels verify candidates against — a wrong word there is
hidden from the mapping tests wherever the exact kernel repairs the result.  Here they are rebuilt on the
host from the gene slices alone, following Indexer::make_index / index_contig (src/core/indexer.rs:122-250:
forward windows 0..len-17, reverse-complement windows 1..len-16, a window with a base other than A/C/G/T
has no key, lower case is upper-cased first) and compared with gf_index_export.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CODE = {65: 0, 67: 1, 84: 2, 71: 3}   # A C T G: (ascii >> 1) & 3; complement = code ^ 2
M32 = 0xFFFFFFFF


def _export(ix, what):
    from genefuserust_amd import _lib
    L, h = _lib.lib(), ix._handle()
    n = L.gf_index_export(h, what, None, 0)
    assert n >= 0
    buf = np.zeros(max(n // 4, 1), dtype=np.uint32)
    assert L.gf_index_export(h, what, buf.ctypes.data, n) == n
    return buf[: n // 4]


def _field_reverse(x):
    r = 0
    for k in range(16):
        r |= ((x >> (2 * k)) & 3) << (2 * (15 - k))
    return r


def _canon14(x):
    r = (_field_reverse(x) >> 4) ^ 0x0AAAAAAA
    return min(r, x)


def _filter_bits(s14, nwords):
    h = (_canon14(s14) * 0x9E3779B1) & M32
    m = h ^ (h >> 15)
    return (h * nwords) >> 32, (1 << (m & 31)) | (1 << ((m >> 5) & 31))


def _genes(rng):
    """Gene slices around every boundary of the build: shorter than a window, exactly one window, one key per
    strand, tile edges (4096 window starts per block), runs of N, lower case, repeats (2..5-fold and
    >= 6-fold keys), a gene that repeats another one's stretch."""
    def rnd(n):
        return bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n))
    g = [rnd(n) for n in (5, 16, 17, 18, 31, 33, 4095, 4096, 4097, 4111, 4113, 9000)]
    g.append(b"")
    a = bytearray(rnd(3000))
    a[100:103] = b"NNN"
    a[1500] = ord("n")
    a[2999] = ord("N")
    a[0] = ord("R")
    g.append(bytes(a))
    g.append(rnd(700).lower())
    unit = rnd(40)
    g.append(rnd(50) + unit * 3 + rnd(60) + unit + rnd(10))          # 2..5-fold keys
    g.append(rnd(20) + b"AC" * 40 + rnd(20) + rnd(30) * 8)           # >= 6-fold keys
    g.append(g[11][1000:1400] + rnd(100))                            # shares 400 bases with another gene
    g.append(b"A" * 64)                                              # its own reverse complement is all T
    g.append(b"ACGT" * 16)                                           # palindromic windows: forward key == reverse key
    return g


def _expected(genes, lin_base, gd_words, filter_words):
    even = np.zeros(gd_words, dtype=np.uint64)
    flags = np.zeros(gd_words, dtype=np.uint64)
    sites = {}   # key (int, base 0 in the low bits) -> list of site codes
    for c, raw in enumerate(genes):
        s = raw.upper()
        n, B = len(s), int(lin_base[c])
        codes = [CODE.get(ch, -1) for ch in s]
        for f, cd in enumerate(codes):
            if cd < 0:
                continue
            p = B + f
            even[p >> 4] |= cd << (2 * (p & 15))
            if f >= 1:
                p = B - f
                even[p >> 4] |= (cd ^ 2) << (2 * (p & 15))
        for f in range(0, n - 15):
            w = codes[f:f + 16]
            if min(w) < 0:
                continue
            key = sum(cd << (2 * k) for k, cd in enumerate(w))
            if f + 16 < n:                 # forward windows 0 .. len-17
                sites.setdefault(key, []).append(B + f)
            if f >= 1:                     # reverse-complement windows
                rkey = _field_reverse(key) ^ 0xAAAAAAAA
                sites.setdefault(rkey, []).append(B - (f + 15))
    filt = np.zeros(max(filter_words, 1), dtype=np.uint64)
    for key, where in sites.items():
        if len(where) == 1:
            p = where[0]
            flags[p >> 4] |= 1 << (2 * (p & 15))
        if len(where) >= 6:                # r04: every site of a key that cannot vote (>= 6 sites) carries the odd bit
            for p in where:
                flags[p >> 4] |= 2 << (2 * (p & 15))
        for s14 in (key & 0x0FFFFFFF, key >> 4):
            w, b = _filter_bits(s14, filter_words)
            filt[w] |= b
    return even.astype(np.uint32), flags.astype(np.uint32), filt.astype(np.uint32), sites


def test_strands_flags_and_filter_word_by_word(gpu_device):
    from genefuserust_amd import Indexer, _lib
    rng = np.random.default_rng(20260)
    genes = _genes(rng)
    ix = Indexer.from_gene_slices(genes)
    ix.make_index()
    try:
        gdu = _export(ix, 0)
        filt = _export(ix, 1)
        lin_base = _export(ix, 2)
        assert lin_base.shape[0] == len(genes) and gdu.shape[0] % 2 == 0
        gd_words = gdu.shape[0] // 2
        even, flags, want_filter, sites = _expected(genes, lin_base, gd_words, filt.shape[0])
        got_even, got_flags = gdu[0::2], gdu[1::2]
        bad = np.nonzero(got_even != even)[0]
        assert bad.size == 0, ("strand words differ", bad[:5], [hex(int(got_even[i])) for i in bad[:5]],
                               [hex(int(even[i])) for i in bad[:5]])
        bad = np.nonzero(got_flags != flags)[0]
        assert bad.size == 0, ("unique flags differ", bad[:5], [hex(int(got_flags[i])) for i in bad[:5]],
                               [hex(int(flags[i])) for i in bad[:5]])
        assert int((flags & 0x55555555).astype(np.uint64).sum()) > 0 and int((flags & 0xAAAAAAAA).astype(np.uint64).sum()) > 0
        assert int((flags & (flags >> 1) & 0x55555555).sum()) == 0    # no site is both the only one of its key and one of six
        # the filter holds exactly the canonical 14-mers of the keys: no missing bit (a false negative would
        # drop reads), no bit more than the build rule gives
        bad = np.nonzero(filt != want_filter)[0]
        assert bad.size == 0, ("filter words differ", bad[:5])
        info = ix.info()
        assert info["n_keys"] == len(sites) and info["n_unique"] == sum(1 for v in sites.values() if len(v) == 1)
        assert info["n_high_keys"] == sum(1 for v in sites.values() if len(v) >= 6) and info["n_high_keys"] > 0
        assert info["n_dupe_keys"] == sum(1 for v in sites.values() if 2 <= len(v) <= 5) and info["n_dupe_keys"] > 0
    finally:
        ix.close()


def test_repeat_heavy_genes_through_the_side_list(gpu_device):
    """Genes that are almost nothing but repeats: nearly every site is a later site of its key, so the one-pass
    build's side list takes them all (block buffers flushed many times per tile, counters bumped by thousands of
    threads at once) — strands, flags, filter and statistics still equal the host rebuild, and every key answers
    with its sites (2..5-fold) or none (>= 6-fold), like the reference's m_dupe_list / the HIGH mark."""
    from genefuserust_amd import Indexer
    rng = np.random.default_rng(99)

    def rnd(n):
        return bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n))
    unit3 = rnd(700)
    genes = [b"ACGTTGCA" * 6000,            # 8 distinct windows per strand, thousands of sites each: all HIGH
             b"A" * 20000,                   # one key (and its reverse complement), 20 K sites each
             unit3 * 3 + rnd(40),            # every key of the unit three times: 2..5-fold lists
             rnd(500) + unit3 + rnd(500),    # ... a fourth time, in another gene
             b"N" * 3000,                    # nothing to index
             rnd(5000)]
    ix = Indexer.from_gene_slices(genes)
    ix.make_index()
    try:
        gdu, filt, lin_base = _export(ix, 0), _export(ix, 1), _export(ix, 2)
        gd_words = gdu.shape[0] // 2
        even, flags, want_filter, sites = _expected(genes, lin_base, gd_words, filt.shape[0])
        assert (gdu[0::2] == even).all() and (gdu[1::2] == flags).all() and (filt == want_filter).all()
        info = ix.info()
        assert info["n_keys"] == len(sites) and info["n_sites"] == sum(len(v) for v in sites.values())
        assert info["n_unique"] == sum(1 for v in sites.values() if len(v) == 1)
        assert info["n_dupe_keys"] == sum(1 for v in sites.values() if 2 <= len(v) <= 5) >= 600
        assert info["n_high_keys"] == sum(1 for v in sites.values() if len(v) >= 6) >= 10
        assert info["n_dupe_sites"] == sum(len(v) for v in sites.values() if 2 <= len(v) <= 5)
        # every key through the look-up: device keys -> reference-coded k-mers is a bijection the library owns,
        # so ask with the windows themselves
        def ref_kmer(key):   # device key (base 0 in the low bits, A0 C1 T2 G3) -> indexer.rs:789-913 coding
            out = 0
            for k in range(16):
                c = (key >> (2 * k)) & 3
                out = (out << 2) | {0: 0, 1: 2, 2: 1, 3: 3}[c]
            return out
        keys = list(sites.keys())
        sample = [keys[i] for i in rng.choice(len(keys), size=min(len(keys), 3000), replace=False)]
        cnt, ctg, pos = ix.lookup(np.array([ref_kmer(k) for k in sample], dtype=np.uint32))
        for j, k in enumerate(sample):
            n = len(sites[k])
            assert int(cnt[j]) == (n if n <= 5 else -2), (hex(k), n, int(cnt[j]))   # -2: the HIGH mark
            if 2 <= n <= 5:   # the list holds exactly the key's sites (as site codes: contig / position -> code)
                got = [int(lin_base[int(ctg[j, t])]) + int(pos[j, t]) for t in range(n)]
                # ... in ascending order of site code: the thread that brings a list's last site sorts it
                # (gf_k_index_side), so the content is reproducible whatever the order of arrival
                assert got == sorted(sites[k]), (hex(k), got, sorted(sites[k]))
    finally:
        ix.close()


def test_filter_partition_overflow_goes_through_the_atomic(gpu_device):
    """The presence filter is filled by hash partitions (gf_k_filter_scatter / gf_k_filter_build); a partition has
    room for twice an even share of the hashes.  120 K bases of one letter put 120 K equal hashes into ONE of some
    25 partitions: the excess takes the atomic path and the building block ORs over it — the filter still equals
    the host rebuild word by word, with the poly-A key's bits in it."""
    from genefuserust_amd import Indexer
    rng = np.random.default_rng(4711)
    def rnd(n):
        return bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n))
    # (+ 150 slices of 20-45 bases: tiles of more than 62 genes take the kernel's search of the offsets per base
    #  instead of the tile's own short list)
    genes = [b"A" * 120000, rnd(90000), b"CA" * 20000] + [rnd(int(n)) for n in rng.integers(20, 46, size=150)]
    ix = Indexer.from_gene_slices(genes)
    ix.make_index()
    try:
        gdu, filt, lin_base = _export(ix, 0), _export(ix, 1), _export(ix, 2)
        assert filt.shape[0] > 4 * 8192   # several partitions
        even, flags, want_filter, sites = _expected(genes, lin_base, gdu.shape[0] // 2, filt.shape[0])
        bad = np.nonzero(filt != want_filter)[0]
        assert bad.size == 0, ("filter words differ", bad[:5])
        assert (gdu[0::2] == even).all() and (gdu[1::2] == flags).all()
        w, b = _filter_bits(0, filt.shape[0])   # the canonical 14-mer of A x 14 (and of T x 14)
        assert int(filt[w]) & b == b
    finally:
        ix.close()


@pytest.mark.parametrize("copies", [2, 5, 6])
def test_a_gene_list_with_every_gene_repeated(gpu_device, copies):
    """The same genes listed 2, 5 and 6 times: every key is 2-fold / 5-fold (every site in a duplicate list — the room
    for the lists is handed out by granules, and here nothing may go to waste: all of dupes[] is spoken for) or 6-fold
    (all HIGH, no list at all).  Strands, flags, filter, statistics and the lists themselves against the host rebuild."""
    from genefuserust_amd import Indexer
    rng = np.random.default_rng(77 + copies)
    base = [bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n)) for n in (30000, 17000, 9000)]
    genes = base * copies
    ix = Indexer.from_gene_slices(genes)
    ix.make_index()
    try:
        gdu, filt, lin_base = _export(ix, 0), _export(ix, 1), _export(ix, 2)
        even, flags, want_filter, sites = _expected(genes, lin_base, gdu.shape[0] // 2, filt.shape[0])
        assert (gdu[0::2] == even).all() and (gdu[1::2] == flags).all() and (filt == want_filter).all()
        assert int((flags & 0x55555555).astype(np.uint64).sum()) == 0   # nobody is the only site of its key
        # ... and with six copies every site is one of six or more: the odd bits (r04) mark them all
        assert (int((flags & 0xAAAAAAAA).astype(np.uint64).sum()) > 0) == (copies >= 6)
        info = ix.info()
        n_sites = sum(len(v) for v in sites.values())
        assert info["n_keys"] == len(sites) and info["n_sites"] == n_sites and info["n_unique"] == 0
        if copies <= 5:
            assert info["n_dupe_keys"] == len(sites) and info["n_dupe_sites"] == n_sites and info["n_high_keys"] == 0
        else:
            assert info["n_dupe_keys"] == 0 and info["n_dupe_sites"] == 0 and info["n_high_keys"] == len(sites)

        def ref_kmer(key):   # device key -> indexer.rs:789-913 coding
            out = 0
            for k in range(16):
                out = (out << 2) | {0: 0, 1: 2, 2: 1, 3: 3}[(key >> (2 * k)) & 3]
            return out
        keys = list(sites.keys())
        sample = [keys[i] for i in rng.choice(len(keys), size=2000, replace=False)]
        cnt, ctg, pos = ix.lookup(np.array([ref_kmer(k) for k in sample], dtype=np.uint32))
        for j, k in enumerate(sample):
            if copies <= 5:
                got = [int(lin_base[int(ctg[j, t])]) + int(pos[j, t]) for t in range(copies)]
                assert int(cnt[j]) == copies and got == sorted(sites[k])
            else:
                assert int(cnt[j]) == -2
    finally:
        ix.close()


def test_export_rejects_unknown_array(gpu_device):
    from genefuserust_amd import Indexer, _lib
    ix = Indexer.from_gene_slices([b"ACGTTGCA" * 8])
    ix.make_index()
    try:
        assert _lib.lib().gf_index_export(ix._handle(), 7, None, 0) < 0
    finally:
        ix.close()


def test_rebuilds_reuse_device_blocks_and_trim_returns_them(gpu_device):
    """Multi-CSV mode frees and rebuilds the index per CSV (fusion_scan.rs:62-188): the freed index's device
    blocks are kept for the next build, rebuilt indexes answer like the first build, gf_index_trim gives the
    blocks back to the device."""
    import torch
    from genefuserust_amd import Indexer, _lib
    rng = np.random.default_rng(7)
    big = [bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=1_500_000)) for _ in range(2)]
    small = _genes(rng)
    first = None
    for rep in range(3):       # alternating sizes, like the CSV list
        for genes in (big, small):
            ix = Indexer.from_gene_slices(genes)
            ix.make_index()
            if genes is small:
                arrays = [_export(ix, k).copy() for k in (0, 1, 2)]
                if first is None:
                    first = arrays
                else:   # a reused (dirty) block is cleared like a fresh one
                    assert all((a == b).all() for a, b in zip(first, arrays))
            table_bytes = ix.info()["table_bytes"]
            ix.close()
    ix = Indexer.from_gene_slices([b"ACGTTGCA" * 8])
    ix.make_index()
    try:
        torch.cuda.synchronize()
        free_a = torch.cuda.mem_get_info(gpu_device)[0]
        _lib.check(_lib.lib().gf_index_trim(ix._handle()))
        free_b = torch.cuda.mem_get_info(gpu_device)[0]
        assert free_b - free_a >= 3_000_000 * 2 * 8 // 4 * 8 // 2, (free_a, free_b)   # at least half of the big table
        assert (_export(ix, 2) == _export(ix, 2)).all()
    finally:
        ix.close()
