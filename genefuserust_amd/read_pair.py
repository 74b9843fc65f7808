"""SURVEY.md §8(f)-2 — the step before the hot path: ``SequenceReadPair::fast_merge``
(src/core/read.rs:313-440) on the device, and the pair policy of
``PairEndScanner::scan_pair_end`` (src/core/pescanner.rs:427-518) over a whole batch.

Device work is ``gf_fast_merge_device`` (csrc/gf_merge_kernels.h) behind the C ABI; torch is
used for device buffers and one prefix sum only.  No CPU fallback: without the HIP library
and a GPU every call here raises.
"""
from __future__ import annotations

import ctypes as C
from typing import List, NamedTuple, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .fusion_mapper import FusionMapper, ReadMatch, reverse_complement
from .indexer import BytesLike, Indexer, _as_bytes


class MergedRead(NamedTuple):
    """The SequenceRead fast_merge returns: name suffix " merged_diff_{diff}" (read.rs:372)."""
    seq: bytes
    quality: bytes
    diff: int


class SequenceReadPair:
    """read.rs:300-311.  ``left``/``right`` are (seq, quality) as read from the FASTQ files."""

    def __init__(self, left: Tuple[BytesLike, BytesLike], right: Tuple[BytesLike, BytesLike]):
        self.m_left = (_as_bytes(left[0]), _as_bytes(left[1]))
        self.m_right = (_as_bytes(right[0]), _as_bytes(right[1]))
        if len(self.m_left[0]) != len(self.m_left[1]) or len(self.m_right[0]) != len(self.m_right[1]):
            raise ValueError("sequence and quality lengths differ")

    def fast_merge(self, indexer: Indexer) -> Optional[MergedRead]:
        """One pair through the C ABI (a one-thread launch; batches use fast_merge_device)."""
        (ls, lq), (rs, rq) = self.m_left, self.m_right
        cap = len(ls) + len(rs) + 1
        oseq, oqual = C.create_string_buffer(cap), C.create_string_buffer(cap)
        olen, odiff = C.c_int32(0), C.c_int32(0)
        rc = _lib.check(_lib.lib().gf_fast_merge(indexer._handle(), ls, lq, len(ls), rs, rq, len(rs), oseq, oqual,
                                                 C.byref(olen), C.byref(odiff)))
        if rc == 0:
            return None
        return MergedRead(oseq.raw[:olen.value], oqual.raw[:olen.value], int(odiff.value))


def pack_reads(seqs: Sequence[bytes]) -> Tuple[np.ndarray, np.ndarray]:
    off = np.zeros(len(seqs) + 1, dtype=np.int64)
    if seqs:
        np.cumsum([len(s) for s in seqs], out=off[1:])
    return np.frombuffer(b"".join(seqs), dtype=np.uint8).copy(), off


def fast_merge_device(indexer: Indexer, l_bases, l_quals, l_off, r_bases, r_quals, r_off, max_read_len: int,
                      stream=None, with_quals: bool = True):
    """Merge a batch of pairs resident in HBM.  Returns (bases, quals, offsets, diff): the
    merged reads packed back to back in pair order — offsets int64[n+1], a pair that does not
    merge has an empty slot — in the layout ``Indexer.map_reads_device`` takes, plus diff int32[n].
    gf_fast_merge_find_device, one prefix sum, gf_fast_merge_write_device.  ``with_quals=False``: the bases
    alone (quals is None) — the form gf_scan_pairs_device uses, a kernel of its own."""
    import torch
    n = l_off.numel() - 1
    dev = l_bases.device
    for t in (l_bases, l_quals, r_bases, r_quals):
        assert t.dtype == torch.uint8 and t.is_cuda
    assert l_off.dtype == torch.int64 and r_off.dtype == torch.int64 and r_off.numel() == n + 1
    st = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
    out_len = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    out_diff = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    L, h = _lib.lib(), indexer._handle()
    args = (l_bases.data_ptr(), l_quals.data_ptr(), l_off.data_ptr(), r_bases.data_ptr(), r_quals.data_ptr(),
            r_off.data_ptr(), n)
    _lib.check(L.gf_fast_merge_find_device(h, *args, int(max_read_len), out_len.data_ptr(), out_diff.data_ptr(), st))
    offsets = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    if n:
        torch.cumsum(out_len[:n], 0, out=offsets[1:])
    total = int(offsets[-1].item()) if n else 0
    bases = torch.empty(max(total, 1), dtype=torch.uint8, device=dev)
    quals = torch.empty(max(total, 1), dtype=torch.uint8, device=dev) if with_quals else None
    _lib.check(L.gf_fast_merge_write_device(h, *args, out_len.data_ptr(), offsets.data_ptr(), bases.data_ptr(),
                                            quals.data_ptr() if with_quals else None, st))
    return bases[:total], (quals[:total] if with_quals else None), offsets, out_diff[:n]


def fast_merge_batch(indexer: Indexer, pairs: Sequence[SequenceReadPair]) -> List[Optional[MergedRead]]:
    """Host convenience over fast_merge_device: upload, merge, download."""
    import torch
    if not pairs:
        return []
    dev = torch.device("cuda", indexer.info()["device"])
    lb, lo = pack_reads([p.m_left[0] for p in pairs])
    lq, _ = pack_reads([p.m_left[1] for p in pairs])
    rb, ro = pack_reads([p.m_right[0] for p in pairs])
    rq, _ = pack_reads([p.m_right[1] for p in pairs])
    t = [torch.from_numpy(a).to(dev) for a in (lb, lq, lo, rb, rq, ro)]
    max_len = int(max(np.diff(lo).max(), np.diff(ro).max()))
    bases, quals, off, diff = fast_merge_device(indexer, *t, max_len)
    torch.cuda.synchronize(dev)
    b, q, o, d = bases.cpu().numpy().tobytes(), quals.cpu().numpy().tobytes(), off.cpu().numpy(), diff.cpu().numpy()
    return [MergedRead(b[o[i]:o[i + 1]], q[o[i]:o[i + 1]], int(d[i])) if o[i + 1] > o[i] else None
            for i in range(len(pairs))]


class PairScan(NamedTuple):
    """What gf_scan_pairs_device leaves in HBM: ``hits`` uint8[cap, 64] (gf_pair_hit records),
    ``bases`` / ``quals`` uint8 (the matched reads, at seq_offset), ``totals`` int64[8]."""
    hits: "object"
    bases: "object"
    quals: "object"
    totals: "object"

    def download(self) -> Tuple[np.ndarray, bytes, bytes, dict]:
        """Synchronises.  (records as PAIR_HIT_DTYPE, bases, quals, totals dict)."""
        t = self.totals.cpu().numpy()
        tot = {"hits": int(t[0]), "hit_bytes": int(t[1]), "merged_pairs": int(t[2]), "retried_reads": int(t[3]),
               "overflow": int(t[4])}
        k, nb = min(tot["hits"], self.hits.shape[0]), min(tot["hit_bytes"], self.bases.numel())
        rec = self.hits[:k].cpu().numpy().view(_lib.PAIR_HIT_DTYPE).reshape(-1)
        return rec, self.bases[:nb].cpu().numpy().tobytes(), self.quals[:nb].cpu().numpy().tobytes(), tot


def scan_pairs_device(indexer: Indexer, l_bases, l_quals, l_off, r_bases, r_quals, r_off, max_read_len: int,
                      pair_id_base: int = 0, hits_cap: Optional[int] = None, bytes_cap: Optional[int] = None,
                      retry_cap: int = 0, stream=None, l_qual_off=None, r_qual_off=None) -> PairScan:
    """``PairEndScanner::scan_pair_end`` (pescanner.rs:427-518) for a pack of pairs resident in HBM,
    one asynchronous call: gf_scan_pairs_device.  No host round trip between merge, the mapping
    passes, the reverse-complement retries and the compaction of the matched reads.

    ``l_qual_off`` / ``r_qual_off`` (both or neither): the qualities were left in the FASTQ texts
    (fastq_cut_device(lean=True)) — ``l_quals`` / ``r_quals`` are the texts and these say where each record's
    quality line starts (gf_scan_pairs_text_device)."""
    import torch
    n = l_off.numel() - 1
    dev = l_bases.device
    for t in (l_bases, l_quals, r_bases, r_quals):
        assert t.dtype == torch.uint8 and t.is_cuda
    assert l_off.dtype == torch.int64 and r_off.dtype == torch.int64 and r_off.numel() == n + 1
    st = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
    hits_cap = max(1024, n // 16) if hits_cap is None else hits_cap
    bytes_cap = hits_cap * 2 * max(int(max_read_len), 1) if bytes_cap is None else bytes_cap
    hits = torch.empty((max(hits_cap, 1), 64), dtype=torch.uint8, device=dev)
    hb = torch.empty(max(bytes_cap, 1), dtype=torch.uint8, device=dev)
    hq = torch.empty(max(bytes_cap, 1), dtype=torch.uint8, device=dev)
    totals = torch.zeros(8, dtype=torch.int64, device=dev)
    assert (l_qual_off is None) == (r_qual_off is None)
    if l_qual_off is not None:
        assert l_qual_off.dtype == torch.int64 and r_qual_off.dtype == torch.int64
        assert l_qual_off.numel() >= n and r_qual_off.numel() >= n
        _lib.check(_lib.lib().gf_scan_pairs_text_device(
            indexer._handle(), l_bases.data_ptr(), l_quals.data_ptr(), l_qual_off.data_ptr(), l_off.data_ptr(), l_bases.numel(),
            r_bases.data_ptr(), r_quals.data_ptr(), r_qual_off.data_ptr(), r_off.data_ptr(), r_bases.numel(), n,
            int(max_read_len), int(pair_id_base), int(retry_cap), hits.data_ptr(), hits_cap, hb.data_ptr(), hq.data_ptr(),
            bytes_cap, totals.data_ptr(), st))
        return PairScan(hits, hb, hq, totals)
    _lib.check(_lib.lib().gf_scan_pairs_device(
        indexer._handle(), l_bases.data_ptr(), l_quals.data_ptr(), l_off.data_ptr(), l_bases.numel(),
        r_bases.data_ptr(), r_quals.data_ptr(), r_off.data_ptr(), r_bases.numel(), n, int(max_read_len),
        int(pair_id_base), int(retry_cap), hits.data_ptr(), hits_cap, hb.data_ptr(), hq.data_ptr(), bytes_cap,
        totals.data_ptr(), st))
    return PairScan(hits, hb, hq, totals)


def finish_pair_hits_device(indexer: Indexer, scan: "PairScan", stream=None):
    """The tail on the device (gf_pair_hits_finish_device): make_match + calc_distance for the records of a pair scan
    while they are still in HBM.  Returns (readmatch uint8[cap, 28] tensor, status int32[cap] tensor); rows beyond
    totals[0] are untouched.  Asynchronous."""
    import torch
    dev = scan.hits.device
    cap = int(scan.hits.shape[0])
    out = torch.zeros((max(cap, 1), _lib.READMATCH_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    status = torch.zeros(max(cap, 1), dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
    _lib.check(_lib.lib().gf_pair_hits_finish_device(indexer._handle(), scan.hits.data_ptr(), scan.totals.data_ptr(), cap,
                                                     scan.bases.data_ptr(), out.data_ptr(), status.data_ptr(), st))
    return out, status


def finish_pair_hits(mapper: FusionMapper, rec: np.ndarray, bases: bytes, quals: bytes,
                     threads: int = 8) -> List[Tuple[int, ReadMatch]]:
    """The host-side tail for the records of a pair scan: FusionMapper::make_match + calc_distance
    (fusion_mapper.rs:154-251) on each matched read — one gf_pair_hits_finish call for the whole
    list (host threads inside the library).  Returns (pair_id, ReadMatch) in push order."""
    from .indexer import GenePos
    n = int(rec.shape[0])
    if n == 0:
        return []
    rec = np.ascontiguousarray(rec)
    rm = np.zeros(n, dtype=_lib.READMATCH_DTYPE)
    status = np.zeros(n, dtype=np.int32)
    _lib.check(_lib.lib().gf_pair_hits_finish(mapper.m_indexer._handle(), rec.ctypes.data, n, bases, len(bases),
                                              rm.ctypes.data, status.ctypes.data, int(threads)))
    src_name = ("merged", "r1", "r2")
    out: List[Tuple[int, ReadMatch]] = []
    for h, r, st in zip(rec, rm, status):
        if st != _lib.GF_RM_MATCH:
            continue
        o, ln = int(h["seq_offset"]), int(h["read_len"])
        src = src_name[int(h["source"])]
        m = ReadMatch(bases[o:o + ln], int(r["read_break"]), GenePos(int(r["left_contig"]), int(r["left_position"])),
                      GenePos(int(r["right_contig"]), int(r["right_position"])), int(r["gap"]), int(r["left_distance"]),
                      int(r["right_distance"]), bool(h["flags"] & 2), b"", src, quals[o:o + ln],
                      int(h["merge_diff"]) if src == "merged" else -1)
        out.append((int(h["pair_id"]), m))
    return out


def scan_pair_end(mapper: FusionMapper, pairs: Sequence[SequenceReadPair]) -> List[List[ReadMatch]]:
    """pescanner.rs:427-518 for a pack of pairs: upload the pack, ONE device call
    (``scan_pairs_device``), download the matched reads, finish them on the host.  Returns, per
    pair, the matches in the order the reference pushes them."""
    import torch
    if not pairs:
        return []
    ix = mapper.m_indexer
    dev = torch.device("cuda", ix.info()["device"])
    lb, lo = pack_reads([p.m_left[0] for p in pairs])
    lq, _ = pack_reads([p.m_left[1] for p in pairs])
    rb, ro = pack_reads([p.m_right[0] for p in pairs])
    rq, _ = pack_reads([p.m_right[1] for p in pairs])
    t = [torch.from_numpy(a).to(dev) for a in (lb, lq, lo, rb, rq, ro)]
    max_len = max(int(np.diff(lo).max()), int(np.diff(ro).max()), 1)
    n = len(pairs)
    res = scan_pairs_device(ix, *t, max_len, hits_cap=3 * n, bytes_cap=int(lb.size + rb.size) * 2 + 64)
    rec, hb, hq, tot = res.download()
    if tot["overflow"] & 1:   # more reverse-complement retries than the default capacity: once more with room for all
        res = scan_pairs_device(ix, *t, max_len, hits_cap=3 * n, bytes_cap=int(lb.size + rb.size) * 2 + 64, retry_cap=3 * n)
        rec, hb, hq, tot = res.download()
    assert not tot["overflow"], tot
    out: List[List[ReadMatch]] = [[] for _ in pairs]
    for p, m in finish_pair_hits(mapper, rec, hb, hq):
        out[p].append(m)
    return out


def scan_pair_end_stepwise(mapper: FusionMapper, pairs: Sequence[SequenceReadPair]) -> List[List[ReadMatch]]:
    """The same policy with the host between the steps (the first form; kept as a second
    implementation the device pipeline is tested against): three GPU steps instead of up to five
    ``map_read`` calls per pair: merge all pairs; map the merged read of every pair that
    merged and R1, R2 of every pair that did not (one batch); map the reverse complement of
    every candidate that was mapable but gave no match (one smaller batch).  Returns, per
    pair, the matches in the order the reference pushes them.  A match found on the reverse
    complement of R1/R2 has ``m_reversed`` set (:489,:511); one found on the reverse
    complement of a merged read has not (:465-468)."""
    merged = fast_merge_batch(mapper.m_indexer, pairs)
    cands: List[bytes] = []
    quals: List[bytes] = []
    owner: List[Tuple[int, str]] = []  # (pair, "merged" | "r1" | "r2")
    for p, (pair, m) in enumerate(zip(pairs, merged)):
        if m is not None:
            cands.append(m.seq)
            quals.append(m.quality)
            owner.append((p, "merged"))
        else:
            cands.append(pair.m_left[0])
            quals.append(pair.m_left[1])
            owner.append((p, "r1"))
            cands.append(pair.m_right[0])
            quals.append(pair.m_right[1])
            owner.append((p, "r2"))
    first = mapper.map_reads(cands)
    found: List[Optional[ReadMatch]] = [m for m, _ in first]
    for i, m in enumerate(found):
        if m is not None:
            m.m_quality = quals[i]
    retry = [i for i, (m, mapable) in enumerate(first) if m is None and mapable]
    if retry:
        second = mapper.map_reads([reverse_complement(cands[i]) for i in retry])
        for i, (m, _) in zip(retry, second):
            if m is not None:
                if owner[i][1] != "merged":
                    m.m_reversed = True
                m.m_quality = quals[i][::-1]  # SequenceRead::reverse_complement reverses the quality (read.rs:243-261)
                found[i] = m
    out: List[List[ReadMatch]] = [[] for _ in pairs]
    for i, m in enumerate(found):
        if m is not None:
            m.m_source = owner[i][1]
            if m.m_source == "merged":
                m.m_merge_diff = merged[owner[i][0]].diff
            out[owner[i][0]].append(m)
    return out
