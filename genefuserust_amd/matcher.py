"""SURVEY.md §8(f)-3 — ``Matcher`` and ``FusionMapper::remove_alignables`` reproduced AS THEY ARE
(src/core/matcher.rs:32-726, src/core/fusion_mapper.rs:488-542), not as they were meant.

The reference's ``make_kmer*`` (matcher.rs:769-885) leave their loop at the first base (a C++
``switch``'s ``break`` inside a Rust ``for``), so a "k-mer" is the code of ONE base (A 0, T 1,
C 2, G 3).  What follows from that, step by step, is what this module computes:

* the "bloom filter" built from the candidate reads (``init_bloom_filter``, :64-88) has at most
  bits 0..3 of byte 0 set: bit c iff some window of some read (or of its reverse complement)
  starts with base c;
* ``index_contig_bytes`` (:227-289) rolls its value with the base AT the window start, so the
  value is at most 3 — and gets past the filter — only at the first valid base after the
  start of a contig or after a base outside ACGT, and wherever all earlier bases of the run
  (up to 15) are ``A``; the site is filed under the code of the base at that position;
* ``map_to_index`` (:388-529) votes with the keys that have at most 50 sites (shifting each site
  by its INDEX in the site list, not by the read offset), and when any vote exists walks the
  read again with an inverted ``contains_key`` test: a valid window whose key is in the index
  is skipped, one whose key is not makes ``get(..).unwrap()`` panic.  So it returns ``None`` or
  panics; it cannot return a match (the mask stays empty, every base counts as a mismatch, and a
  sequence that passed the first loop has at least 15).

On a human genome every key has far more than 50 sites (each poly-A run and each N gap end adds
some), nothing votes, ``do_match`` is ``None`` for every read and ``remove_alignables`` removes
nothing after scanning the whole genome.  On a small reference it can panic; that is raised here
as ``MatcherPanic``.  Host code, numpy only; parity unpinned (the reference has no test for it):
checked against a literal loop-by-loop model in ``oracle/indexer_model.py``.
"""
from __future__ import annotations

from typing import Dict, List, Mapping, NamedTuple, Optional, Sequence, Tuple

import numpy as np

from .fusion_mapper import ReadMatch, reverse_complement
from .indexer import BytesLike, GenePos, _as_bytes

KMER = 16
SKIP_THRESHOLD = 50
_CODE = np.full(256, -1, dtype=np.int8)
for _b, _c in ((65, 0), (84, 1), (67, 2), (71, 3)):  # A T C G, upper case only (base2num_bytes, :728-747)
    _CODE[_b] = _c


class MatcherPanic(RuntimeError):
    """The reference panics here (``Option::unwrap()`` on ``None``, matcher.rs:494)."""


class MatchResult(NamedTuple):
    start_gp: GenePos
    mismatches: List[int]
    reversed: bool


def _gp_to_i64(contig: int, position: int) -> int:  # :896-903
    return (contig << 32) + position


class Matcher:
    """matcher.rs:32-62."""

    def __init__(self, reference: Optional[Mapping[str, BytesLike]], seqs: Sequence[BytesLike]):
        self.m_reference = reference
        self.m_contig_names: List[str] = []
        self.m_kmer_positions: Dict[int, List[Tuple[int, int]]] = {}
        self.bloom_bits = 0  # bits 0..3 of byte 0: all the reference's 512 MB array can ever hold
        for s in seqs:
            s = _as_bytes(s)
            self._init_bloom_filter_with_seq(s)
            self._init_bloom_filter_with_seq(reverse_complement(s))
        self.make_index()

    from_ref_and_seqs = classmethod(lambda cls, reference, seqs: cls(reference, seqs))

    def _init_bloom_filter_with_seq(self, s: bytes) -> None:  # :73-88
        n = len(s) - KMER + 1
        if n < 0:  # `0..(s.len() - KMER + 1)` underflows in the reference
            raise MatcherPanic("sequence shorter than 15 bases")
        codes = _CODE[np.frombuffer(s, dtype=np.uint8)[:n]]
        for c in range(4):
            if (codes == c).any():
                self.bloom_bits |= 1 << c

    def make_index(self) -> None:  # :120-169, index_contig_bytes :227-289
        if self.m_reference is None:
            return
        for ctg, (name, seq) in enumerate(sorted(self.m_reference.items())):  # m_all_contigs is a BTreeMap
            self.m_contig_names.append(name)
            s = np.frombuffer(_as_bytes(seq).upper(), dtype=np.uint8)
            n = len(s) - KMER
            if n < 0:
                raise MatcherPanic("contig shorter than 16 bases")  # seq.get(..(len - KMER)).unwrap()
            codes = _CODE[s[:n]].astype(np.int64)
            ok = codes >= 0
            idx = np.arange(n)
            # start of the current run of valid bases, and of the current run of A inside it
            run_start = np.maximum.accumulate(np.where(~ok, idx + 1, 0))
            not_a = codes != 0
            a_start = np.maximum.accumulate(np.where(not_a, idx + 1, 0))  # first index of the A's just before i ...
            prev_non_a = np.concatenate(([0], a_start[:-1])) if n else a_start  # ... seen from position i
            since = idx - np.maximum(prev_non_a, run_start)       # bases before i in the run that are all A
            in_run = idx - run_start                              # bases before i in the run
            # the rolled 32-bit value is <= 3 iff the (up to 15) earlier bases still inside it are all A
            rec = ok & (since >= np.minimum(in_run, KMER - 1))
            for c in range(4):
                if not (self.bloom_bits >> c) & 1:
                    continue
                pos = idx[rec & (codes == c)]
                if len(pos):
                    self.m_kmer_positions.setdefault(c, []).extend((ctg, int(p)) for p in pos)

    def map_to_index(self, seq: bytes) -> Optional[MatchResult]:  # :388-529
        n = len(seq)
        nwin = n - KMER + 1
        if nwin < 0:
            raise MatcherPanic("sequence shorter than 15 bases")
        codes = _CODE[np.frombuffer(seq, dtype=np.uint8)[:nwin]]
        kmer_stat: Dict[int, int] = {0: 0}
        for c in codes:
            if c < 0:
                continue
            sites = self.m_kmer_positions.get(int(c))
            if sites is None:
                kmer_stat[0] += 1
            elif len(sites) > SKIP_THRESHOLD:
                pass  # skipped
            else:
                for k, (ctg, pos) in enumerate(sites):  # (shifted by the site's index: the reference's shadowed `i`)
                    g = _gp_to_i64(ctg, pos - k)
                    kmer_stat[g] = kmer_stat.get(g, 0) + 1
        top = sorted(((cnt, g) for g, cnt in kmer_stat.items() if g != 0 and cnt > 0), reverse=True)
        if not top:
            return None
        # some diagonal has a vote: the mask walk starts, and its `contains_key` test is inverted
        for c in codes:
            if c >= 0 and int(c) not in self.m_kmer_positions:
                raise MatcherPanic("called `Option::unwrap()` on a `None` value")
        # every valid window was skipped: the mask stays empty, every base is a mismatch, and a
        # sequence that got this far has at least 15 of them: never fewer than the 10 it would take
        return None

    def do_match(self, seq: BytesLike) -> Optional[MatchResult]:  # :662-689
        s = _as_bytes(seq)
        a = self.map_to_index(s)
        b = self.map_to_index(reverse_complement(s))
        if b is not None:
            b = b._replace(reversed=True)
        if a is None:
            return b
        if b is None:
            return a
        return a if len(a.mismatches) <= len(b.mismatches) else b


def remove_alignables(matches: Sequence[ReadMatch], reference: Optional[Mapping[str, BytesLike]]):
    """fusion_mapper.rs:488-542: (kept, removed).  Scans the whole reference, as the reference does."""
    if reference is None:
        return list(matches), 0
    m = Matcher(reference, [x.m_read for x in matches])
    kept = [x for x in matches if m.do_match(x.m_read) is None]
    return kept, len(matches) - len(kept)
