"""The boundary under load (needs a GPU): concurrent callers, the small-call route, the streaming
entry, gf_index_trim.  C++ threads call the C ABI directly (tests/cpp/test_threads.cpp); the Python
side checks the streaming entry against the one-shot host call on pinned and pageable buffers."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_threads_on_one_index(gpu_device, tmp_path):
    """T = 8 threads on one index through gf_map_reads / gf_map_reads_hits / gf_map_read / gf_stream_*:
    every result equals the serial one (pescanner.rs:296-311 calls map_read from t-1 threads)."""
    exe = str(tmp_path / "test_threads")
    subprocess.run(["g++", "-std=c++17", "-O1", "-pthread", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "test_threads.cpp"), "-o", exe,
                    "-L" + os.path.join(ROOT, "genefuserust_amd"), "-lgfmatch",
                    "-Wl,-rpath," + os.path.join(ROOT, "genefuserust_amd")], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(out.stdout)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout + out.stderr
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "test_threads.log"), "w") as f:
        f.write(out.stdout)


@pytest.mark.parametrize("pinned", [True, False])
def test_stream_equals_one_shot(gpu_device, pinned):
    from genefuserust_amd import Indexer, synth
    from genefuserust_amd.stream import MapStream, pinned_empty
    genes = synth.make_geneset("IDX-T", scale=0.1)
    ix = Indexer.from_gene_slices(genes.seqs, genes.reversed_flags)
    ix.make_index()
    synth.MIXES["TEST"] = (0.2, 0.5, 0.3)
    n, L = 200_003, 150
    rb = synth.make_reads(genes, n, read_len=L, mix="TEST", seed=12)
    bases, offsets = rb.bases.numpy(), rb.offsets.numpy()
    want = ix.map_reads_hits(bases, offsets, read_id_base=5)
    assert want.shape[0] > 20_000
    if pinned:
        hb, ho = pinned_empty(bases.size, np.uint8), pinned_empty(offsets.size, np.int64)
        hb[:], ho[:] = bases, offsets
    else:
        hb, ho = bases, offsets
    pack = 30_000
    got = []
    with MapStream(ix, max_reads=pack, max_bytes=pack * L, depth=3) as ms:
        inflight = 0
        for p0 in range(0, n, pack):
            p1 = min(n, p0 + pack)
            if inflight == ms.depth:
                got.append(ms.collect())
                inflight -= 1
            ms.submit(hb, ho[p0:p1 + 1], read_id_base=5 + p0)
            inflight += 1
        while inflight:
            got.append(ms.collect())
            inflight -= 1
        # an empty pack, and a pack over the stream's capacity
        ms.submit(hb, ho[:1])
        assert ms.collect().shape[0] == 0
        from genefuserust_amd import _lib
        with pytest.raises(_lib.GfError) as e:
            ms.submit(hb, ho[: pack + 2])
        assert e.value.code == _lib.GF_ERR_CAPACITY
    got = np.concatenate(got)
    assert got.tobytes() == want.tobytes()
    ix.close()


def test_host_api_indexes_leave_no_workspace_behind(gpu_device):
    """every host-buffer call maps on a lane's stream and gives that stream a workspace in the
    process-wide pool; freeing the index must take it along.  Build, map through the host API and free many
    indexes (multi-CSV mode with host buffers): the device's free memory must not creep."""
    import torch
    from genefuserust_amd import Indexer, synth
    genes = synth.make_geneset("IDX-T", scale=0.05)
    rb = synth.make_reads(genes, 300_000, read_len=150, mix="PANEL", seed=5)
    bases, offsets = rb.bases.numpy(), rb.offsets.numpy()

    def one():
        ix = Indexer.from_gene_slices(genes.seqs, genes.reversed_flags)
        ix.make_index()
        h = ix.map_reads_hits(bases, offsets)
        ix.close()
        return h.shape[0]

    first = one()
    one()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(gpu_device)[0]
    for _ in range(12):
        assert one() == first
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info(gpu_device)[0]
    # a leaked workspace is 64 B x 300 K reads = 19 MB per index: 12 of them would be 230 MB
    assert free0 - free1 < 32 << 20, (free0, free1)


def test_index_free_with_an_open_stream_is_refused(gpu_device, capfd):
    """gf_stream keeps a pointer to its index: gf_index_free under an open stream leaks the index (with a message)
    instead of leaving the stream dangling; after gf_stream_close the index is freed normally."""
    from genefuserust_amd import Indexer, synth, _lib
    from genefuserust_amd.stream import MapStream
    genes = synth.make_geneset("IDX-T", scale=0.02)
    ix = Indexer.from_gene_slices(genes.seqs, genes.reversed_flags)
    ix.make_index()
    rb = synth.make_reads(genes, 2000, read_len=150, mix="PANEL", seed=6)
    ms = MapStream(ix, max_reads=4000, max_bytes=4000 * 150, depth=2)
    h = ix._handle()
    _lib.lib().gf_index_free(h)          # refused: the stream is open
    assert b"still open" in _lib.lib().gf_last_error()
    ms.submit(rb.bases.numpy(), rb.offsets.numpy())
    assert ms.collect().shape[0] >= 0    # the index is still there
    ms.close()
    ix.close()


def test_pack_route_equals_batch_route(gpu_device, oracle):
    """A host call of up to 8192 reads (a pack of the reference's size, common.rs:23: 1000 pairs) takes the zero-copy
    route — the pack in pinned memory, one launch of the wave-per-read kernels, a completion word instead of a stream
    synchronisation.  Same bytes out as the batch route (copies + flat pipeline) and as the oracle: dense counts,
    hit records, the streaming entry; pageable sources and sources inside a gf_host_alloc block (read in place, at
    every byte phase); packs of 1, 65, 2000, 8192 and 8193 reads; a pack with reads of every length class."""
    from genefuserust_amd import Indexer, synth
    from genefuserust_amd.stream import MapStream, pinned_empty
    genes = synth.make_geneset("IDX-T", scale=0.05)
    ix = Indexer.from_gene_slices(genes.seqs, genes.reversed_flags)
    ix.make_index()
    ox = oracle.OracleIndexer(genes.seqs)
    synth.MIXES["TEST"] = (0.2, 0.5, 0.3)
    n, L = 20_000, 150
    rb = synth.make_reads(genes, n, read_len=L, mix="TEST", seed=41)
    bases, offsets = rb.bases.numpy(), rb.offsets.numpy()
    ocounts, omatches = ox.map_reads_packed(bases, offsets, threads=8)

    def both_routes(fn):
        out = []
        for lim in (0, -1):
            ix.set_pack_call_reads(lim)
            out.append(fn())
        ix.set_pack_call_reads(-1)
        return out

    for phase in (0, 1, 5, 15):
        hb = pinned_empty(bases.size + 16, np.uint8)
        hb[phase:phase + bases.size] = bases
        for src, off in ((bases, offsets), (hb, offsets + phase)):
            for p0, m in ((0, 1), (3, 65), (100, 2000), (4000, 8192), (9000, 8193)):
                o = off[p0:p0 + m + 1]
                (c0, m0), (c1, m1) = both_routes(lambda: ix.map_reads_packed(src, o))
                assert (c0 == c1).all() and (c0 == ocounts[p0:p0 + m]).all()
                nz = c0 > 0
                assert (m0[nz, 0] == m1[nz, 0]).all() and (m1[nz, 0] == omatches[p0:p0 + m][nz, 0]).all()
                two = c0 == 2
                assert (m0[two, 1] == m1[two, 1]).all() and (m1[two, 1] == omatches[p0:p0 + m][two, 1]).all()
                h0, h1 = both_routes(lambda: ix.map_reads_hits(src, o, read_id_base=77 + p0))
                assert h0.tobytes() == h1.tobytes() and h0.shape[0] == int(nz.sum())
                # more hits than the caller's capacity: the total is reported, the first `cap` records written
                if m == 2000:
                    k0, k1 = both_routes(lambda: ix.map_reads_hits(src, o, read_id_base=77 + p0, cap=10))
                    assert k0.tobytes() == k1.tobytes() == h0[:10].tobytes()
        if phase > 1:
            continue
        # the streaming entry: packs of 2000 reads (zero-copy slots) against one batch call
        want = ix.map_reads_hits(bases, offsets, read_id_base=5)
        for src, off in ((bases, offsets), (hb, offsets + phase)):
            def run():
                got = []
                with MapStream(ix, max_reads=2000, max_bytes=2000 * L, depth=3) as ms:
                    inflight = 0
                    for q0 in range(0, n, 2000):
                        if inflight == ms.depth:
                            got.append(ms.collect())
                            inflight -= 1
                        ms.submit(src, off[q0:q0 + 2001], read_id_base=5 + q0)
                        inflight += 1
                    while inflight:
                        got.append(ms.collect())
                        inflight -= 1
                return np.concatenate(got)
            g0, g1 = both_routes(run)
            assert g0.tobytes() == g1.tobytes() == want.tobytes()
    # every length class in one pack (the wave-per-read kernels of the 1024- and 4096-base classes finish the call)
    rng = np.random.default_rng(9)
    g0s, g1s = genes.seqs[0], genes.seqs[1]
    reads = []
    for k in range(300):  # junction reads: the left part from one gene, the right part from another
        ln = int(rng.choice([40, 150, 251, 300, 700, 1024, 1500, 4000]))
        cut = ln // 2 + int(rng.integers(-ln // 8, ln // 8 + 1))
        p, q = int(rng.integers(0, len(g0s) - ln)), int(rng.integers(0, len(g1s) - ln))
        reads.append(g0s[p:p + cut] + g1s[q:q + ln - cut])
    mb, mo = synth.ragged_batch(reads)
    (c0, m0), (c1, m1) = both_routes(lambda: ix.map_reads_packed(mb, mo))
    oc, om = ox.map_reads_packed(mb, mo, threads=4)
    assert (c0 == c1).all() and (c1 == oc).all() and int((oc > 0).sum()) > 100
    nz = oc > 0
    assert (m0[nz, 0] == m1[nz, 0]).all() and (m1[nz, 0] == om[nz, 0]).all()
    ix.close()
