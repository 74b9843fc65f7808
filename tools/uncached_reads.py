"""Experiment: how much of the mapping pass is the reads' bases passing through the L2?  The same batch mapped from
an ordinary device buffer and from one allocated uncached (hipExtMallocWithFlags, hipDeviceMallocUncached), whose
loads do not allocate in the L2.   python3 tools/uncached_reads.py [n_reads]"""
import ctypes as C
import sys
import time

sys.path.insert(0, "/root/repo")
import torch
from genefuserust_amd import Indexer, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
L = 150
genes = synth.make_geneset("IDX-D")
ix = Indexer.from_gene_slices(genes.seqs, genes.reversed_flags)
ix.make_index()
rb = synth.make_reads(genes, n, read_len=L, mix="PANEL", seed=1, device="cuda")
hip = C.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]


class Raw:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def alloc(flags, nbytes):
    p = C.c_void_p()
    rc = hip.hipExtMallocWithFlags(C.byref(p), nbytes, flags)
    assert rc == 0, rc
    return p.value


def run(tag, bases):
    counts = torch.empty(n, dtype=torch.uint8, device="cuda")
    matches = torch.empty((n, 2, 4), dtype=torch.int32, device="cuda")
    for _ in range(2):
        ix.map_reads_device(bases, rb.offsets, L, counts, matches)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        ix.map_reads_device(bases, rb.offsets, L, counts, matches)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 100
    print("%-28s %.3f ms per pass, %.2f G reads/s, reads with segments %d" % (tag, ms, n / ms / 1e6, int((counts > 0).sum())), flush=True)
    return counts.clone()


nbytes = rb.bases.numel()
c0 = run("torch buffer", rb.bases)
for tag, flags in (("hipDeviceMallocDefault", 0x0), ("hipDeviceMallocFinegrained", 0x1), ("hipDeviceMallocUncached", 0x3)):
    try:
        p = alloc(flags, nbytes + 64)
        assert hip.hipMemcpy(p, rb.bases.data_ptr(), nbytes, 3) == 0
        t = torch.as_tensor(Raw(p, nbytes), device="cuda")
        c = run(tag, t)
        assert bool((c == c0).all())
    except Exception as e:  # noqa
        print(tag, "failed:", repr(e))
