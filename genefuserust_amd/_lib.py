"""ctypes binding of libgfmatch.so (the C ABI of include/gfmatch.h).

The shared object is built in-tree by ``__graft_entry__.build()`` /
``genefuserust_amd/csrc/Makefile``.  There is no fallback of any kind: if the
library is missing, importing the compute API raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GFMATCH_LIB lets experiments (tools/ablate.sh) load an alternative build of the same ABI
LIB_PATH = os.environ.get("GFMATCH_LIB") or os.path.join(_HERE, "libgfmatch.so")

GF_OK = 0
GF_ERR_ARG = -1
GF_ERR_HIP = -2
GF_ERR_NO_DEVICE = -3
GF_ERR_CAPACITY = -4
GF_ERR_READ_TOO_LONG = -5
GF_ERR_COMM = -6
GF_COMM_ID_BYTES = 128
GF_MAX_READ_LEN = 4096
GF_COUNT_TOO_LONG = 255


class GfSeqMatch(C.Structure):
    _fields_ = [("seq_start", C.c_int32), ("seq_end", C.c_int32), ("position", C.c_int32),
                ("contig", C.c_int16), ("pad", C.c_int16)]


class GfHit(C.Structure):
    _fields_ = [("read_id", C.c_int64), ("n", C.c_int32), ("pad", C.c_int32), ("m", GfSeqMatch * 2)]


class GfPairHit(C.Structure):
    _fields_ = [("pair_id", C.c_int64), ("source", C.c_int32), ("flags", C.c_int32), ("read_len", C.c_int32),
                ("merge_diff", C.c_int32), ("seq_offset", C.c_int64), ("m", GfSeqMatch * 2)]


class GfReadMatch(C.Structure):
    _fields_ = [("read_break", C.c_int32), ("gap", C.c_int32), ("left_distance", C.c_int32),
                ("right_distance", C.c_int32), ("left_position", C.c_int32), ("right_position", C.c_int32),
                ("left_contig", C.c_int16), ("right_contig", C.c_int16)]


GF_RM_NONE, GF_RM_NONE_MAPABLE, GF_RM_MATCH = 0, 1, 2


class GfOptions(C.Structure):
    _fields_ = [("device", C.c_int32), ("reserved", C.c_int32 * 7)]


class GfIndexInfo(C.Structure):
    _fields_ = [(k, C.c_int64) for k in (
        "n_genes", "total_bp", "n_sites", "n_keys", "n_unique", "n_dupe_keys", "n_high_keys",
        "n_dupe_sites", "n_buckets", "table_bytes")] + [("device", C.c_int32), ("pad", C.c_int32)]


SEQMATCH_DTYPE = np.dtype([("seq_start", "<i4"), ("seq_end", "<i4"), ("position", "<i4"),
                           ("contig", "<i2"), ("pad", "<i2")])
HIT_DTYPE = np.dtype([("read_id", "<i8"), ("n", "<i4"), ("pad", "<i4"), ("m", SEQMATCH_DTYPE, (2,))])
READMATCH_DTYPE = np.dtype([("read_break", "<i4"), ("gap", "<i4"), ("left_distance", "<i4"), ("right_distance", "<i4"),
                            ("left_position", "<i4"), ("right_position", "<i4"), ("left_contig", "<i2"),
                            ("right_contig", "<i2")])
PAIR_HIT_DTYPE = np.dtype([("pair_id", "<i8"), ("source", "<i4"), ("flags", "<i4"), ("read_len", "<i4"),
                           ("merge_diff", "<i4"), ("seq_offset", "<i8"), ("m", SEQMATCH_DTYPE, (2,))])
assert SEQMATCH_DTYPE.itemsize == 16 and HIT_DTYPE.itemsize == 48 and PAIR_HIT_DTYPE.itemsize == 64


class GfError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("gfmatch error %d: %s" % (code, msg))
        self.code = code


_lib = None


def lib() -> C.CDLL:
    """Load libgfmatch.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libgfmatch.so not found at %s — build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950). gfmatch has no CPU fallback." % LIB_PATH)
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7.  If
    # libgfmatch.so were loaded first it would pull /opt/rocm's copy in, torch would
    # then load its own, and the second runtime to initialise finds no device.
    # Importing torch first makes the loader resolve our NEEDED libamdhip64.so.7 to
    # the copy already in the process.  (A C/C++/Rust host without torch simply uses
    # the system runtime.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    L.gf_last_error.restype = C.c_char_p
    L.gf_version.restype = C.c_char_p
    L.gf_index_build.argtypes = [C.POINTER(C.c_char_p), C.POINTER(i64), i32, C.POINTER(GfOptions), C.POINTER(vp)]
    L.gf_index_build.restype = C.c_int
    L.gf_index_free.argtypes = [vp]
    L.gf_index_free.restype = None
    L.gf_index_info_get.argtypes = [vp, C.POINTER(GfIndexInfo)]
    L.gf_index_info_get.restype = C.c_int
    L.gf_index_fusion_seq.argtypes = [vp, i32, C.c_char_p, i64]
    L.gf_index_fusion_seq.restype = i64
    L.gf_index_lookup.argtypes = [vp, vp, i64, vp, vp, vp]
    L.gf_index_lookup.restype = C.c_int
    L.gf_map_reads.argtypes = [vp, vp, vp, i64, vp, vp]
    L.gf_map_reads.restype = C.c_int
    L.gf_map_read.argtypes = [vp, C.c_char_p, i64, C.POINTER(GfSeqMatch)]
    L.gf_map_read.restype = C.c_int
    L.gf_map_reads_hits.argtypes = [vp, vp, vp, i64, i64, vp, i64, C.POINTER(i64)]
    L.gf_map_reads_hits.restype = C.c_int
    L.gf_map_reads_device.argtypes = [vp, vp, vp, i64, i32, vp, vp, vp]
    L.gf_map_reads_device.restype = C.c_int
    L.gf_map_reads_fixed_device.argtypes = [vp, vp, i64, i32, vp, vp, vp]
    L.gf_map_reads_fixed_device.restype = C.c_int
    L.gf_packed_chunks.argtypes = [i64]
    L.gf_packed_chunks.restype = i64
    L.gf_pack_bases_device.argtypes = [vp, vp, i64, vp, vp, vp]
    L.gf_pack_bases_device.restype = C.c_int
    L.gf_pack_bases_host.argtypes = [vp, i64, vp, vp, i32]
    L.gf_pack_bases_host.restype = C.c_int
    L.gf_stream_submit_packed.argtypes = [vp, vp, vp, vp, i64, i64]
    L.gf_stream_submit_packed.restype = C.c_int
    L.gf_map_reads_packed_device.argtypes = [vp, vp, vp, vp, i64, i32, vp, vp, vp]
    L.gf_map_reads_packed_device.restype = C.c_int
    L.gf_compact_workspace_bytes.argtypes = [i64]
    L.gf_compact_workspace_bytes.restype = i64
    L.gf_compact_hits_device.argtypes = [vp, vp, vp, i64, i64, vp, i64, vp, vp, vp]
    L.gf_compact_hits_device.restype = C.c_int
    L.gf_in_required_direction.argtypes = [C.POINTER(GfSeqMatch), i32, vp, i32]
    L.gf_in_required_direction.restype = C.c_int
    L.gf_fusion_map_read.argtypes = [C.POINTER(C.c_char_p), C.POINTER(i64), i32, vp, C.c_char_p, i64,
                                     C.POINTER(GfSeqMatch), i32, C.POINTER(GfReadMatch)]
    L.gf_fusion_map_read.restype = C.c_int
    L.gf_index_fusion_map_read.argtypes = [vp, vp, C.c_char_p, i64, C.POINTER(GfSeqMatch), i32, C.POINTER(GfReadMatch)]
    L.gf_index_fusion_map_read.restype = C.c_int
    L.gf_readmatch_filter.argtypes = [vp, C.c_char_p, i64, i32]
    L.gf_readmatch_filter.restype = C.c_int
    L.gf_readmatch_order.argtypes = [i32, i64, C.c_char_p, i64, i32, i64, C.c_char_p, i64]
    L.gf_readmatch_order.restype = C.c_int
    L.gf_fastq_workspace_bytes.argtypes = [i64]
    L.gf_fastq_workspace_bytes.restype = i64
    L.gf_fastq_index_device.argtypes = [vp, vp, i64, vp, i64, vp, vp, vp]
    L.gf_fastq_index_device.restype = C.c_int
    L.gf_fastq_gather_device.argtypes = [vp, vp, i64, vp, i64, i64, vp, vp, vp, i64, vp, vp, vp]
    L.gf_fastq_gather_device.restype = C.c_int
    L.gf_fastq_gather_lean_device.argtypes = [vp, vp, i64, vp, i64, i64, vp, vp, i64, vp, vp, vp, vp]
    L.gf_fastq_gather_lean_device.restype = C.c_int
    L.gf_fast_merge_find_device.argtypes = [vp] * 7 + [i64, i32] + [vp] * 3
    L.gf_fast_merge_find_device.restype = C.c_int
    L.gf_fast_merge_write_device.argtypes = [vp] * 7 + [i64] + [vp] * 5
    L.gf_fast_merge_write_device.restype = C.c_int
    L.gf_fast_merge_device.argtypes = [vp] * 7 + [i64, i32] + [vp] * 6
    L.gf_fast_merge_device.restype = C.c_int
    L.gf_fast_merge.argtypes = [vp, C.c_char_p, C.c_char_p, i32, C.c_char_p, C.c_char_p, i32, C.c_char_p, C.c_char_p,
                                C.POINTER(i32), C.POINTER(i32)]
    L.gf_fast_merge.restype = C.c_int
    L.gf_index_set_gene_reversed.argtypes = [vp, vp, i32]
    L.gf_index_set_gene_reversed.restype = C.c_int
    L.gf_pair_hits_finish.argtypes = [vp, vp, i64, vp, i64, vp, vp, i32]
    L.gf_pair_hits_finish.restype = C.c_int
    L.gf_pair_hits_finish_device.argtypes = [vp, vp, vp, i64, vp, vp, vp, vp]
    L.gf_pair_hits_finish_device.restype = C.c_int
    L.gf_scan_pairs_retry_capacity.argtypes = [i64]
    L.gf_scan_pairs_retry_capacity.restype = i64
    L.gf_scan_pairs_device.argtypes = [vp, vp, vp, vp, i64, vp, vp, vp, i64, i64, i32, i64, i64, vp, i64, vp, vp, i64, vp, vp]
    L.gf_scan_pairs_device.restype = C.c_int
    L.gf_scan_pairs_text_device.argtypes = [vp, vp, vp, vp, vp, i64, vp, vp, vp, vp, i64, i64, i32, i64, i64, vp, i64, vp, vp, i64, vp, vp]
    L.gf_scan_pairs_text_device.restype = C.c_int
    L.gf_segment_mask_test.argtypes = [vp, vp, vp, i64, vp, vp, vp, vp]
    L.gf_segment_mask_test.restype = C.c_int
    L.gf_index_export.argtypes = [vp, i32, vp, i64]
    L.gf_index_export.restype = i64
    L.gf_index_trim.argtypes = [vp]
    L.gf_index_trim.restype = C.c_int
    L.gf_stream_open.argtypes = [vp, i64, i64, i32, C.POINTER(vp)]
    L.gf_stream_open.restype = C.c_int
    L.gf_stream_submit.argtypes = [vp, vp, vp, i64, i64]
    L.gf_stream_submit.restype = C.c_int
    L.gf_stream_collect.argtypes = [vp, vp, i64, C.POINTER(i64)]
    L.gf_stream_collect.restype = C.c_int
    L.gf_stream_close.argtypes = [vp]
    L.gf_stream_close.restype = None
    L.gf_copy_from_host_device.argtypes = [vp, vp, vp, i64, vp]
    L.gf_copy_from_host_device.restype = C.c_int
    L.gf_host_alloc.argtypes = [i64]
    L.gf_host_alloc.restype = vp
    L.gf_host_free.argtypes = [vp]
    L.gf_host_free.restype = None
    L.gf_edit_distance.argtypes = [C.c_char_p, i64, C.c_char_p, i64]
    L.gf_edit_distance.restype = i64
    L.gf_comm_unique_id.argtypes = [vp]
    L.gf_comm_unique_id.restype = C.c_int
    L.gf_comm_init.argtypes = [vp, i32, i32, i32, C.POINTER(vp)]
    L.gf_comm_init.restype = C.c_int
    L.gf_comm_rank.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.gf_comm_rank.restype = C.c_int
    L.gf_comm_free.argtypes = [vp]
    L.gf_comm_free.restype = None
    L.gf_allgather_workspace_bytes.argtypes = [i32, i64]
    L.gf_allgather_workspace_bytes.restype = i64
    L.gf_allgather_hits_device.argtypes = [vp, vp, vp, i64, vp, vp, vp, vp]
    L.gf_allgather_hits_device.restype = C.c_int
    L.gf_pack_gathered_hits_device.argtypes = [vp, i32, i64, vp, vp, vp]
    L.gf_pack_gathered_hits_device.restype = C.c_int
    L.gf_set_profiling.argtypes = [vp, i32]
    L.gf_set_profiling.restype = C.c_int
    L.gf_last_stage_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.gf_last_stage_ms.restype = C.c_int
    L.gf_set_map_variant.argtypes = [vp, i32]
    L.gf_set_map_variant.restype = C.c_int
    L.gf_set_pack_call_reads.argtypes = [vp, C.c_int64]
    L.gf_set_pack_call_reads.restype = C.c_int
    L.gf_last_map_kernel_ms.argtypes = [vp]
    L.gf_last_map_kernel_ms.restype = C.c_float
    _lib = L
    return L


def check(rc: int) -> int:
    if rc < 0:
        raise GfError(rc, lib().gf_last_error().decode("utf-8", "replace"))
    return rc
