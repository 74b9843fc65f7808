"""Streaming host entry (gf_stream_* of include/gfmatch.h): packs of reads submitted ahead of the
ones being mapped, hit records collected in submission order.  The reference's consumer threads
take packs off a queue while the producer reads the FASTQ (pescanner.rs:255-311); here the queue
is on the device side of the link, so the copy of pack k+1 overlaps the kernels of pack k.
No CPU fallback: every call goes through libgfmatch.so."""
from __future__ import annotations

import ctypes as C
import weakref

import numpy as np

from . import _lib
from ._lib import HIT_DTYPE
from .indexer import Indexer


def pinned_empty(n: int, dtype) -> np.ndarray:
    """A numpy array over pinned host memory (gf_host_alloc): buffers that cross the link
    asynchronously.  Freed when the array (and every view of it) is gone."""
    dt = np.dtype(dtype)
    nbytes = max(int(n) * dt.itemsize, 1)
    L = _lib.lib()
    p = L.gf_host_alloc(nbytes)
    if not p:
        raise MemoryError("gf_host_alloc(%d) failed: %s" % (nbytes, L.gf_last_error().decode()))
    buf = (C.c_uint8 * nbytes).from_address(p)
    arr = np.frombuffer(buf, dtype=dt, count=int(n))
    weakref.finalize(buf, L.gf_host_free, p)
    return arr


class MapStream:
    def __init__(self, indexer: Indexer, max_reads: int, max_bytes: int, depth: int = 2):
        self._h = C.c_void_p()
        self.depth = int(depth)
        self.max_reads = int(max_reads)
        _lib.check(_lib.lib().gf_stream_open(indexer._handle(), int(max_reads), int(max_bytes), int(depth), C.byref(self._h)))
        self._ix = indexer            # the stream must not outlive its index
        self._hits = np.empty(self.max_reads, dtype=HIT_DTYPE)   # a pack cannot return more hits than reads

    def submit(self, bases: np.ndarray, offsets: np.ndarray, read_id_base: int = 0) -> None:
        """``offsets`` int64[n+1] are positions in ``bases`` (absolute, like gf_map_reads): a pack may
        be a slice of the offsets of one large buffer."""
        assert bases.dtype == np.uint8 and offsets.dtype == np.int64 and offsets.flags.c_contiguous
        n = offsets.size - 1
        _lib.check(_lib.lib().gf_stream_submit(self._h, bases.ctypes.data, offsets.ctypes.data, n, int(read_id_base)))

    def submit_packed(self, pk: np.ndarray, iv: np.ndarray, offsets: np.ndarray, read_id_base: int = 0) -> None:
        """The same pack in packed form (``pack_bases_host``): 0.375 bytes per base over the link instead of 1."""
        assert pk.dtype == np.uint32 and iv.dtype == np.uint16 and offsets.dtype == np.int64 and offsets.flags.c_contiguous
        n = offsets.size - 1
        _lib.check(_lib.lib().gf_stream_submit_packed(self._h, pk.ctypes.data, iv.ctypes.data, offsets.ctypes.data, n,
                                                      int(read_id_base)))

    def collect(self) -> np.ndarray:
        """Hit records (HIT_DTYPE) of the oldest pack in flight."""
        total = C.c_int64(0)
        _lib.check(_lib.lib().gf_stream_collect(self._h, self._hits.ctypes.data, self._hits.size, C.byref(total)))
        k = int(total.value)
        if k > self._hits.size:   # a pack of mostly hits: the caller's buffer was too small, which cannot be repaired after the fact
            raise _lib.GfError(_lib.GF_ERR_CAPACITY, "pack returned %d hits, collect buffer holds %d" % (k, self._hits.size))
        return self._hits[:k].copy()

    def close(self) -> None:
        if self._h:
            _lib.lib().gf_stream_close(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pack_bases_host(bases: np.ndarray, threads: int = 1, pinned: bool = False):
    """gf_pack_bases_host: (pk uint32[chunks], iv uint16[chunks]) of an ASCII base buffer, on the host (no device
    needed); ``pinned`` puts the result in pinned memory so that ``submit_packed`` copies asynchronously."""
    assert bases.dtype == np.uint8 and bases.flags.c_contiguous
    chunks = int(_lib.lib().gf_packed_chunks(bases.size))
    pk = pinned_empty(chunks, np.uint32) if pinned else np.empty(chunks, dtype=np.uint32)
    iv = pinned_empty(chunks, np.uint16) if pinned else np.empty(chunks, dtype=np.uint16)
    _lib.check(_lib.lib().gf_pack_bases_host(bases.ctypes.data, bases.size, pk.ctypes.data, iv.ctypes.data, int(threads)))
    return pk, iv
