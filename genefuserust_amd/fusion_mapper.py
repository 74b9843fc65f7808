"""Host-side mirror of the part of ``FusionMapper`` that sits directly on the hot path
(src/core/fusion_mapper.rs:93-251): ``map_read`` = Indexer.map_read,
the ``mapable`` rule, the direction gate, ``make_match`` and ``calc_distance``.

Same names and behaviour as the Rust type; the batch forms put every read through one
GPU call (``Indexer.map_reads_packed``) and only the reads that come back with two
segments (about 0.1 %) through the host logic, like the reference keeps that logic on
the CPU.  ``scan_single_end`` adds the reverse-complement retry of the scanners
(sescanner.rs:188-195, pescanner.rs:458-513).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import GF_RM_MATCH, GF_RM_NONE, GfReadMatch, GfSeqMatch
from .indexer import BytesLike, GenePos, Indexer, SeqMatch, _as_bytes

_COMP = bytes.maketrans(b"ACGTacgt", b"TGCATGCA")


def reverse_complement(seq: bytes) -> bytes:
    """sequence.rs:22-60: anything outside ACGTacgt becomes N, output upper case."""
    clean = bytes(c if c in b"ACGTacgt" else 78 for c in seq)
    return clean.translate(_COMP)[::-1]


@dataclass
class ReadMatch:
    """The fields of src/core/read_match.rs:18-30 that make_match / calc_distance set."""
    m_read: bytes
    m_read_break: int
    m_left_gp: GenePos
    m_right_gp: GenePos
    m_gap: int
    m_left_distance: int
    m_right_distance: int
    m_reversed: bool = False
    m_name: bytes = b""   # SequenceRead.m_name: the last key of the sort order (read_match.rs:226)
    m_source: str = ""    # which read of a pair it was found on: "merged", "r1" or "r2" (scan_pair_end)
    m_quality: bytes = b""  # SequenceRead.m_quality of m_read (reversed when m_read is a reverse complement)
    m_merge_diff: int = -1  # the N of the " merged_diff_N" name suffix when m_read is a merged pair (read.rs:372)


def edit_distance(a: BytesLike, b: BytesLike) -> int:
    a, b = _as_bytes(a), _as_bytes(b)
    return int(_lib.check(_lib.lib().gf_edit_distance(a, len(a), b, len(b))))


class FusionMapper:
    def __init__(self, indexer: Indexer):
        self.m_indexer = indexer
        self._rev = np.array([f.is_reversed() for f in indexer.m_fusions] or [0], dtype=np.uint8)

    def _tail(self, seq: bytes, mapping: Sequence[SeqMatch]) -> Tuple[Optional[ReadMatch], bool]:
        n = len(mapping)
        arr = (GfSeqMatch * max(n, 1))()
        for k, m in enumerate(mapping):
            arr[k] = GfSeqMatch(m.seq_start, m.seq_end, m.start_gp.position, m.start_gp.contig, 0)
        out = GfReadMatch()
        st = _lib.check(_lib.lib().gf_index_fusion_map_read(self.m_indexer._handle(), self._rev.ctypes.data, seq,
                                                            len(seq), arr, n, C.byref(out)))
        if st != GF_RM_MATCH:
            return None, st != GF_RM_NONE
        return ReadMatch(seq, out.read_break, GenePos(out.left_contig, out.left_position),
                         GenePos(out.right_contig, out.right_position), out.gap, out.left_distance,
                         out.right_distance), True

    def map_read(self, r: BytesLike) -> Tuple[Optional[ReadMatch], bool]:
        """FusionMapper::map_read: returns (match or None, mapable)."""
        seq = _as_bytes(r)
        return self._tail(seq, self.m_indexer.map_read(seq))

    def map_reads(self, reads: Sequence[BytesLike]) -> List[Tuple[Optional[ReadMatch], bool]]:
        from .synth import ragged_batch
        seqs = [_as_bytes(r) for r in reads]
        bases, offsets = ragged_batch(seqs)
        counts, matches = self.m_indexer.map_reads_packed(bases, offsets)
        out: List[Tuple[Optional[ReadMatch], bool]] = [(None, False)] * len(seqs)
        for r in np.nonzero(counts >= 2)[0]:
            mp = [SeqMatch(int(matches[r, k]["seq_start"]), int(matches[r, k]["seq_end"]),
                           GenePos(int(matches[r, k]["contig"]), int(matches[r, k]["position"])))
                  for k in range(int(counts[r]))]
            out[int(r)] = self._tail(seqs[int(r)], mp)
        return out

    @staticmethod
    def _c_readmatch(m: ReadMatch) -> GfReadMatch:
        return GfReadMatch(m.m_read_break, m.m_gap, m.m_left_distance, m.m_right_distance, m.m_left_gp.position,
                           m.m_right_gp.position, m.m_left_gp.contig, m.m_right_gp.contig)

    def filter_matches(self, matches: Sequence[ReadMatch], deletion_threshold: int = 50):
        """FusionMapper::filter_matches without remove_alignables (fusion_mapper.rs:276-376):
        (kept, {"complexity": n, "distance": n, "indels": n}), the reference's three counters."""
        kept: List[ReadMatch] = []
        removed = {"complexity": 0, "distance": 0, "indels": 0}
        names = (None, "complexity", "distance", "indels")
        for m in matches:
            seq = _as_bytes(m.m_read)
            rm = self._c_readmatch(m)
            why = _lib.check(_lib.lib().gf_readmatch_filter(C.byref(rm), seq, len(seq), int(deletion_threshold)))
            if why == 0:
                kept.append(m)
            else:
                removed[names[why]] += 1
        return kept, removed

    def remove_alignables(self, matches: Sequence[ReadMatch]):
        """fusion_mapper.rs:488-542, the last step of ``filter_matches``, as the reference behaves:
        (kept, removed).  See matcher.py — it removes nothing, or panics on a small reference."""
        from .matcher import remove_alignables
        return remove_alignables(matches, self.m_indexer.get_ref())

    @staticmethod
    def sort_matches(matches: Sequence[ReadMatch]) -> List[ReadMatch]:
        """fusion_mapper.rs:378-384: read_break descending, shorter read first, name descending."""
        import functools
        L = _lib.lib()

        def cmp(a: ReadMatch, b: ReadMatch) -> int:
            return L.gf_readmatch_order(a.m_read_break, len(a.m_read), a.m_name, len(a.m_name), b.m_read_break,
                                        len(b.m_read), b.m_name, len(b.m_name))
        return sorted(matches, key=functools.cmp_to_key(cmp))

    def scan_single_end(self, reads: Sequence[BytesLike]) -> List[Optional[ReadMatch]]:
        """sescanner.rs:188-195: map the read; when it is mapable but gives no match, map its
        reverse complement (set_reversed(true) on that match).  Two GPU calls per batch."""
        first = self.map_reads(reads)
        result: List[Optional[ReadMatch]] = [m for m, _ in first]
        retry = [i for i, (m, mapable) in enumerate(first) if m is None and mapable]
        if retry:
            second = self.map_reads([reverse_complement(_as_bytes(reads[i])) for i in retry])
            for i, (m, _) in zip(retry, second):
                if m is not None:
                    m.m_reversed = True
                    result[i] = m
        return result
