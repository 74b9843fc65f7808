"""gf_pack_bases_host (the host half of the packed hand-over): the AVX2 / plain C++ packer against a numpy model of
gf_k_pack_bases on the CPU; on the GPU against the device kernel itself, and the packed streaming entry against the
ASCII one."""
import numpy as np
import pytest

from genefuserust_amd import _lib


def _model(b: np.ndarray):
    chunks = int(_lib.lib().gf_packed_chunks(b.size))
    buf = np.zeros(chunks * 16, dtype=np.uint8)
    buf[: b.size] = b
    ok = np.isin(buf, np.frombuffer(b"ACGT", dtype=np.uint8))
    ok[b.size:] = False
    code = ((buf >> 1) & 3).astype(np.uint32).reshape(chunks, 16)
    pk = (code << (2 * np.arange(16, dtype=np.uint32))[None, :]).sum(axis=1).astype(np.uint32)
    iv = ((~ok).reshape(chunks, 16).astype(np.uint32) << np.arange(16, dtype=np.uint32)[None, :]).sum(axis=1).astype(np.uint16)
    return pk, iv


@pytest.mark.parametrize("n", [0, 1, 15, 16, 17, 31, 32, 33, 47, 48, 1000, 65536 * 16 * 3 + 5])
def test_host_packer_equals_the_model(n):
    from genefuserust_amd.stream import pack_bases_host
    rng = np.random.default_rng(n)
    b = rng.choice(np.frombuffer(b"ACGTACGTACGTNacgtn@[`{\x00\xff", dtype=np.uint8), size=n).astype(np.uint8)
    for threads in (1, 5):
        pk, iv = pack_bases_host(b, threads=threads)
        wpk, wiv = _model(b)
        assert np.array_equal(pk, wpk) and np.array_equal(iv, wiv)


def test_host_packer_every_byte_value():
    from genefuserust_amd.stream import pack_bases_host
    b = np.tile(np.arange(256, dtype=np.uint8), 3)
    pk, iv = pack_bases_host(b)
    wpk, wiv = _model(b)
    assert np.array_equal(pk, wpk) and np.array_equal(iv, wiv)


@pytest.mark.gpu
def test_host_packer_equals_the_device_kernel_and_packed_stream_equals_ascii(gpu_device):
    import torch
    from genefuserust_amd import Indexer, synth
    from genefuserust_amd.stream import MapStream, pack_bases_host, pinned_empty
    genes = synth.make_geneset("IDX-T", scale=0.1)
    ix = Indexer.from_gene_slices(genes.seqs, genes.reversed_flags)
    ix.make_index()
    synth.MIXES["TEST"] = (0.2, 0.5, 0.3)
    n, L = 150_001, 150
    rb = synth.make_reads(genes, n, read_len=L, mix="TEST", seed=21)
    bases, offsets = rb.bases.numpy(), rb.offsets.numpy()
    bases[12345] = ord("n")
    pk, iv = pack_bases_host(bases, threads=4, pinned=True)
    dpk, div = ix.pack_bases_device(torch.from_numpy(bases).cuda())
    assert np.array_equal(pk, dpk.cpu().numpy().view(np.uint32)) and np.array_equal(iv, div.cpu().numpy().view(np.uint16))
    want = ix.map_reads_hits(bases, offsets, read_id_base=3)
    assert want.shape[0] > 10_000
    pack = 40_000
    got = []
    ho = pinned_empty(offsets.size, np.int64)
    ho[:] = offsets
    with MapStream(ix, max_reads=pack, max_bytes=pack * L, depth=2) as ms:
        inflight = 0
        for p0 in range(0, n, pack):
            p1 = min(n, p0 + pack)
            if inflight == ms.depth:
                got.append(ms.collect())
                inflight -= 1
            ms.submit_packed(pk, iv, ho[p0:p1 + 1], read_id_base=3 + p0)
            inflight += 1
        while inflight:
            got.append(ms.collect())
            inflight -= 1
        ms.submit_packed(pk, iv, ho[:1])     # an empty pack
        assert ms.collect().shape[0] == 0
    got = np.concatenate(got)
    assert got.tobytes() == want.tobytes()
    ix.close()


@pytest.mark.gpu
def test_packed_stream_with_long_reads_and_unaligned_first_offset(gpu_device, oracle):
    """gf_stream_submit_packed with reads of 300 and 1000 bases (the wave-per-read list kernels' packed staging, which
    reads past a read's last chunk: every packed buffer carries four chunks beyond its bases) in packs whose
    first offset is no multiple of 16: same hit records as the ASCII one-shot call and as the oracle."""
    from genefuserust_amd import Indexer, synth
    from genefuserust_amd.stream import MapStream, pack_bases_host, pinned_empty
    genes = synth.make_geneset("IDX-T", scale=0.05)
    ix = Indexer.from_gene_slices(genes.seqs, genes.reversed_flags)
    ix.make_index()
    ox = oracle.OracleIndexer(genes.seqs)
    rng = np.random.default_rng(17)
    g0, g1 = genes.seqs[0], genes.seqs[1]
    reads = []
    for k in range(1200):
        ln = int(rng.choice([150, 300, 300, 1000, 77, 331]))
        cut = ln // 2 + int(rng.integers(-ln // 8, ln // 8 + 1))
        p, q = int(rng.integers(0, len(g0) - ln)), int(rng.integers(0, len(g1) - ln))
        reads.append(g0[p:p + cut] + g1[q:q + ln - cut] if k % 3 else bytes(rng.choice(list(b"ACGT"), size=ln).astype(np.uint8)))
    bases, offsets = synth.ragged_batch([b"ACGTACG"] + reads)    # a 7-base read in front: every later offset is odd
    pk, iv = pack_bases_host(bases, threads=2, pinned=True)
    ho = pinned_empty(offsets.size, np.int64)
    ho[:] = offsets
    n = offsets.size - 1
    want = ix.map_reads_hits(bases, offsets, read_id_base=0)
    oc, om = ox.map_reads_packed(bases, offsets, threads=4)
    assert want.shape[0] == int((oc > 0).sum()) > 200
    got = []
    pack = 173
    with MapStream(ix, max_reads=pack, max_bytes=pack * 1000, depth=3) as ms:
        inflight = 0
        for p0 in range(1, n, pack):        # packs start at read 1: offsets[0] of the first pack is 7
            p1 = min(n, p0 + pack)
            if inflight == ms.depth:
                got.append(ms.collect())
                inflight -= 1
            ms.submit_packed(pk, iv, ho[p0:p1 + 1], read_id_base=p0)
            inflight += 1
        while inflight:
            got.append(ms.collect())
            inflight -= 1
    got = np.concatenate(got)
    assert got.tobytes() == want.tobytes()   # (read 0, seven bases, has no hit)
    ix.close()
