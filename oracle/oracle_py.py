"""ORACLE loader — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.

ctypes wrapper around oracle/liboracle.so (the C++ restatement in
indexer_oracle.cc).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product package never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")

ORC_SEQMATCH = np.dtype([("seq_start", "<i4"), ("seq_end", "<i4"), ("position", "<i4"),
                         ("contig", "<i2"), ("pad", "<i2")])

_lib = None


def build() -> None:
    subprocess.run(["make", "-C", _HERE, "liboracle.so"], check=True, stdout=subprocess.DEVNULL)


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    src = os.path.join(_HERE, "indexer_oracle.cc")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        build()
    L = C.CDLL(_SO)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    L.orc_index_build.argtypes = [C.POINTER(C.c_char_p), C.POINTER(i64), i32]
    L.orc_index_build.restype = vp
    L.orc_index_free.argtypes = [vp]
    L.orc_index_free.restype = None
    L.orc_index_stats.argtypes = [vp, C.POINTER(i64)]
    L.orc_index_keys.argtypes = [vp, vp, i64]
    L.orc_index_keys.restype = i64
    L.orc_index_lookup.argtypes = [vp, i64, C.POINTER(C.c_int16), C.POINTER(i32)]
    L.orc_index_lookup.restype = i32
    L.orc_index_fusion_seq.argtypes = [vp, i32, C.c_char_p, i64]
    L.orc_index_fusion_seq.restype = i64
    L.orc_map_read.argtypes = [vp, C.c_char_p, i64, vp]
    L.orc_map_read.restype = i32
    L.orc_map_reads.argtypes = [vp, vp, vp, i64, i32, vp, vp]
    L.orc_map_reads.restype = None
    L.orc_make_kmer.argtypes = [C.c_char_p, i32, i64, i32]
    L.orc_make_kmer.restype = i64
    L.orc_gp_to_i64.argtypes = [C.c_int16, i32]
    L.orc_gp_to_i64.restype = i64
    L.orc_i64_to_gp.argtypes = [i64, C.POINTER(C.c_int16), C.POINTER(i32)]
    L.orc_segment_mask.argtypes = [vp, i32, C.c_int16, i32, C.c_int16, i32, vp]
    L.orc_segment_mask.restype = i32
    L.orc_reverse_complement.argtypes = [C.c_char_p, i64, C.c_char_p]
    L.orc_in_required_direction.argtypes = [vp, i32, vp]
    L.orc_in_required_direction.restype = i32
    L.orc_edit_distance.argtypes = [C.c_char_p, i64, C.c_char_p, i64]
    L.orc_edit_distance.restype = i64
    L.orc_fusion_map_read.argtypes = [vp, vp, C.c_char_p, i64, vp, i32, vp]
    L.orc_fusion_map_read.restype = i32
    L.orc_fast_merge.argtypes = [C.c_char_p, C.c_char_p, i32, C.c_char_p, C.c_char_p, i32, C.c_char_p, C.c_char_p,
                                 C.POINTER(i32), C.POINTER(i32)]
    L.orc_fast_merge.restype = i32
    L.orc_fastq_cut.argtypes = [C.c_char_p, i64, vp, i64]
    L.orc_fastq_cut.restype = i64
    _lib = L
    return L


Match = Tuple[int, int, int, int]  # (seq_start, seq_end, contig, position)


class OracleIndexer:
    def __init__(self, slices: Sequence[Optional[bytes]]):
        L = lib()
        n = len(slices)
        keep = [s if s is not None else b"" for s in slices]
        arr = (C.c_char_p * max(n, 1))(*keep) if n else (C.c_char_p * 1)()
        lens = (C.c_int64 * max(n, 1))(*[(-1 if s is None else len(s)) for s in slices])
        self._h = C.c_void_p(L.orc_index_build(arr, lens, n))
        self.n_genes = n

    def close(self):
        if self._h:
            lib().orc_index_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stats(self) -> dict:
        out = (C.c_int64 * 4)()
        lib().orc_index_stats(self._h, out)
        return {"m_unique_pos": out[0], "m_dupe_pos": out[1], "n_keys": out[2], "n_high_keys": out[3]}

    def keys(self) -> np.ndarray:
        n = lib().orc_index_keys(self._h, None, 0)
        a = np.zeros(max(n, 1), dtype=np.int64)
        lib().orc_index_keys(self._h, a.ctypes.data, n)
        return a[:n]

    def lookup(self, kmer: int):
        c = (C.c_int16 * 5)()
        p = (C.c_int32 * 5)()
        n = lib().orc_index_lookup(self._h, int(kmer), c, p)
        if n <= 0:
            return n, []
        return n, sorted((int(c[k]), int(p[k])) for k in range(n))

    def fusion_seq(self, c: int) -> str:
        ln = lib().orc_index_fusion_seq(self._h, c, None, 0)
        buf = C.create_string_buffer(max(int(ln), 1))
        lib().orc_index_fusion_seq(self._h, c, buf, ln)
        return buf.raw[:ln].decode("latin-1")

    def map_read(self, seq: bytes) -> List[Match]:
        out = np.zeros(2, dtype=ORC_SEQMATCH)
        n = lib().orc_map_read(self._h, seq, len(seq), out.ctypes.data)
        return [(int(out[k]["seq_start"]), int(out[k]["seq_end"]), int(out[k]["contig"]),
                 int(out[k]["position"])) for k in range(n)]

    def map_reads_packed(self, bases: np.ndarray, offsets: np.ndarray, threads: int = 1):
        b = np.ascontiguousarray(bases, dtype=np.uint8)
        o = np.ascontiguousarray(offsets, dtype=np.int64)
        n = o.size - 1
        counts = np.zeros(max(n, 0), dtype=np.int32)
        matches = np.zeros((max(n, 0), 2), dtype=ORC_SEQMATCH)
        lib().orc_map_reads(self._h, b.ctypes.data if b.size else None, o.ctypes.data, n, threads,
                            counts.ctypes.data, matches.ctypes.data)
        return counts, matches


def make_kmer(seq: bytes, pos: int, last: int = -1, step: int = 1) -> int:
    return int(lib().orc_make_kmer(seq, pos, last, step))


def gp_to_i64(contig: int, position: int) -> int:
    return int(lib().orc_gp_to_i64(contig, position))


def i64_to_gp(v: int) -> Tuple[int, int]:
    c = C.c_int16()
    p = C.c_int32()
    lib().orc_i64_to_gp(v, C.byref(c), C.byref(p))
    return int(c.value), int(p.value)


def segment_mask(mask: Sequence[int], gp1: Tuple[int, int], gp2: Tuple[int, int]) -> List[Match]:
    m = np.asarray(mask, dtype=np.uint8)
    out = np.zeros(2, dtype=ORC_SEQMATCH)
    n = lib().orc_segment_mask(m.ctypes.data, m.size, gp1[0], gp1[1], gp2[0], gp2[1], out.ctypes.data)
    return [(int(out[k]["seq_start"]), int(out[k]["seq_end"]), int(out[k]["contig"]), int(out[k]["position"]))
            for k in range(n)]


def reverse_complement(s: bytes) -> bytes:
    buf = C.create_string_buffer(max(len(s), 1))
    lib().orc_reverse_complement(s, len(s), buf)
    return buf.raw[:len(s)]


def in_required_direction(matches: Sequence[Match], reversed_flags: Sequence[bool]) -> bool:
    arr = np.zeros(max(len(matches), 1), dtype=ORC_SEQMATCH)
    for k, m in enumerate(matches):
        arr[k] = (m[0], m[1], m[3], m[2], 0)
    rev = np.asarray(list(reversed_flags) or [0], dtype=np.uint8)
    return bool(lib().orc_in_required_direction(arr.ctypes.data, len(matches), rev.ctypes.data))


ORC_READMATCH = np.dtype([("read_break", "<i4"), ("gap", "<i4"), ("left_distance", "<i4"), ("right_distance", "<i4"),
                          ("left_position", "<i4"), ("right_position", "<i4"), ("left_contig", "<i2"),
                          ("right_contig", "<i2")])


def edit_distance(a: bytes, b: bytes) -> int:
    return int(lib().orc_edit_distance(a, len(a), b, len(b)))


def fusion_map_read(ox: "OracleIndexer", reversed_flags: Sequence[bool], seq: bytes, mapping: Sequence[Match]):
    """FusionMapper::map_read after Indexer::map_read: (status, readmatch dict or None);
    status 0 = None/unmapable, 1 = None/mapable, 2 = match."""
    arr = np.zeros(max(len(mapping), 1), dtype=ORC_SEQMATCH)
    for k, m in enumerate(mapping):
        arr[k] = (m[0], m[1], m[3], m[2], 0)
    rev = np.asarray(list(reversed_flags) or [0], dtype=np.uint8)
    out = np.zeros(1, dtype=ORC_READMATCH)
    st = int(lib().orc_fusion_map_read(ox._h, rev.ctypes.data, seq, len(seq), arr.ctypes.data, len(mapping),
                                       out.ctypes.data))
    return st, ({k: int(out[0][k]) for k in ORC_READMATCH.names} if st == 2 else None)


def fast_merge(l_seq: bytes, l_qual: bytes, r_seq: bytes, r_qual: bytes):
    """SequenceReadPair::fast_merge: (merged_seq, merged_qual, diff) or None."""
    cap = len(l_seq) + len(r_seq) + 1
    oseq, oqual = C.create_string_buffer(cap), C.create_string_buffer(cap)
    olen, odiff = C.c_int32(0), C.c_int32(0)
    ok = lib().orc_fast_merge(l_seq, l_qual, len(l_seq), r_seq, r_qual, len(r_seq), oseq, oqual, C.byref(olen),
                              C.byref(odiff))
    if not ok:
        return None
    return oseq.raw[:olen.value], oqual.raw[:olen.value], int(odiff.value)


def fastq_cut(text: bytes):
    """FastqReader::read until it returns None: list of (name, sequence, strand, quality)."""
    n = lib().orc_fastq_cut(text, len(text), None, 0)
    out = np.zeros((max(n, 1), 8), dtype=np.int64)
    lib().orc_fastq_cut(text, len(text), out.ctypes.data, n)
    return [tuple(text[out[i, 2 * k]:out[i, 2 * k] + out[i, 2 * k + 1]] for k in range(4)) for i in range(n)]
