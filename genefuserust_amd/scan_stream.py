"""The paired-end scan as a stream of FASTQ text: ``PairEndScanner::scan`` reads both files in
packs while its consumers map them (src/core/pescanner.rs:255-395); here the host hands over raw
text chunks — any byte boundaries, it does not parse — and everything from the records to the hit
list happens on the device, chunk k+1 crossing the link while chunk k is processed:

    host text (pinned)  --H2D, copy stream-->  device text buffer (behind the carried-over tail
    of the previous chunk)  -->  gf_fastq_index_device / gf_fastq_gather_device  -->
    gf_scan_pairs_device (merge, map, reverse-complement retries, ordered compaction)  -->
    gf_pair_hit records + their reads, back to the host; gf_pair_hits_finish there.

A chunk ends anywhere; the bytes after its last complete record (of the record count both files
share) are carried to the front of the next chunk on the device.  Two tiny read-backs per chunk
(the line counts, the carry positions) are the only synchronisation.  No CPU fallback.
"""
from __future__ import annotations

from typing import Iterator, List, Optional, Tuple

import numpy as np

from . import _lib
from .fusion_mapper import FusionMapper, ReadMatch
from .indexer import Indexer
from .read_pair import finish_pair_hits, scan_pairs_device

CARRY_MAX = 1 << 20  # bytes kept in front of a chunk for the previous chunk's tail


class _Side:
    """One FASTQ text: a host byte source and two device buffers the chunks alternate between."""

    def __init__(self, text: np.ndarray, chunk_bytes: int, dev):
        import torch
        assert text.dtype == np.uint8 and text.ndim == 1
        self.text = text
        self.pos = 0
        self.bufs = [torch.empty(CARRY_MAX + chunk_bytes + 64, dtype=torch.uint8, device=dev) for _ in range(2)]
        self.carry_len = 0
        self.chunk_len = [0, 0]

    def done(self) -> bool:
        return self.pos >= self.text.size

    def upload(self, slot: int, nbytes: int, stream) -> None:
        """Queue the H2D copy of the next ``nbytes`` of text into buffer ``slot`` (after the carry area)."""
        import torch
        n = int(max(0, min(nbytes, self.text.size - self.pos)))
        self.chunk_len[slot] = n
        if n:
            # in pieces: the small read-backs of the chunk being processed (line counts, carry positions, hit
            # records) share the copy engines with this upload and would otherwise wait behind all of it
            piece = 8 << 20
            with torch.cuda.stream(stream):
                for o in range(0, n, piece):
                    k = min(piece, n - o)
                    src = torch.from_numpy(self.text[self.pos + o:self.pos + o + k])
                    self.bufs[slot][CARRY_MAX + o:CARRY_MAX + o + k].copy_(src, non_blocking=True)
        self.pos += n


class _Peek:
    """Small values back from the device by kernel stores into pinned memory (gf_copy_to_host_device): a
    hipMemcpy read-back queues behind the upload in flight on the copy engines and waits for all of it."""

    def __init__(self, indexer: Indexer, dev, slots: int = 64, bulk_bytes: int = 16 << 20):
        import torch
        self.h = indexer._handle()
        self.dev = dev
        self.small = torch.empty(2 * slots, dtype=torch.int64).pin_memory()      # 16 bytes per slot
        self.bulk = torch.empty(bulk_bytes, dtype=torch.uint8).pin_memory()
        self.n = 0
        self.pending = []   # (slot, element inside its 16-byte vector)

    def i64(self, t, index: int) -> int:
        """Queue the read-back of int64 element ``index`` of device tensor ``t``; returns a ticket."""
        import torch
        addr = t.data_ptr() + 8 * int(index)
        base = addr & ~15
        st = torch.cuda.current_stream(self.dev).cuda_stream
        _lib.check(_lib.lib().gf_copy_to_host_device(self.h, base, self.small.data_ptr() + 16 * self.n, 16, st))
        self.pending.append((self.n, (addr - base) // 8))
        self.n += 1
        return len(self.pending) - 1

    def wait(self) -> List[int]:
        """Synchronise the stream once; the values of all tickets, in order."""
        import torch
        torch.cuda.current_stream(self.dev).synchronize()
        out = [int(self.small[2 * s + e]) for s, e in self.pending]
        self.pending, self.n = [], 0
        return out

    def bytes(self, t, nbytes: int) -> Optional[bytes]:
        """The first ``nbytes`` of device tensor ``t`` (synchronises); None when it does not fit the bulk block."""
        import torch
        if nbytes == 0:
            return b""
        if nbytes + 16 > self.bulk.numel() or t.data_ptr() & 15 or t.numel() * t.element_size() < ((nbytes + 15) & ~15):
            return None
        st = torch.cuda.current_stream(self.dev).cuda_stream
        _lib.check(_lib.lib().gf_copy_to_host_device(self.h, t.data_ptr(), self.bulk.data_ptr(), nbytes, st))
        torch.cuda.current_stream(self.dev).synchronize()
        return self.bulk[:nbytes].numpy().tobytes()


def scan_pair_text_stream(indexer: Indexer, r1_text: np.ndarray, r2_text: np.ndarray, chunk_bytes: int = 128 << 20,
                          max_read_len: int = 320) -> Iterator[Tuple[np.ndarray, bytes, bytes, dict]]:
    """Yields, per chunk, what ``PairScan.download`` returns — (gf_pair_hit records with pair ids counted
    from the start of the files, the matched reads' bases, their qualities, totals) — for the FASTQ
    texts ``r1_text`` / ``r2_text`` (uint8 arrays; pinned memory makes the copies asynchronous).
    Records pair up by position; the shorter file ends both (fastq_reader.rs:209-218)."""
    import torch
    dev = torch.device("cuda", indexer.info()["device"])
    copy_stream = torch.cuda.Stream(dev)
    main = torch.cuda.current_stream(dev)
    sides = [_Side(r1_text, chunk_bytes, dev), _Side(r2_text, chunk_bytes, dev)]
    L, h = _lib.lib(), indexer._handle()
    peek = _Peek(indexer, dev)
    ready = [None, None]   # per slot: event after which the slot's copies have landed
    free = [None, None]    # per slot: event after which the slot's buffers may be overwritten
    cap_text = CARRY_MAX + chunk_bytes + 64
    st_main = main.cuda_stream
    # per side: the buffers of the record cut, allocated once
    cut = []
    for _ in sides:
        cut.append(dict(
            ws=torch.empty(int(L.gf_fastq_workspace_bytes(cap_text)), dtype=torch.uint8, device=dev),
            n_lines=torch.zeros(2, dtype=torch.int64, device=dev),
            nl_pos=torch.empty(cap_text // 8 + 1024, dtype=torch.int64, device=dev),
            offsets=torch.zeros(cap_text // 8 + 2, dtype=torch.int64, device=dev),
            bases=torch.empty(cap_text, dtype=torch.uint8, device=dev),
            quals=torch.empty(cap_text, dtype=torch.uint8, device=dev),
            n_bad=torch.zeros(2, dtype=torch.int64, device=dev)))

    def start_upload(slot: int):
        if free[slot] is not None:
            copy_stream.wait_event(free[slot])
        for s in sides:   # the file that is ahead (longer carry) gets fewer new bytes: the carries stay bounded
            s.upload(slot, chunk_bytes - s.carry_len, copy_stream)
        ev = torch.cuda.Event()
        ev.record(copy_stream)
        ready[slot] = ev

    pairs_done = 0
    slot = 0
    carries = [torch.empty(0, dtype=torch.uint8, device=dev), torch.empty(0, dtype=torch.uint8, device=dev)]
    start_upload(0)
    while True:
        final = [s.done() for s in sides]   # (after this slot's upload was queued)
        if not all(final):
            start_upload(slot ^ 1)   # next chunk's copy overlaps this chunk's kernels
        main.wait_event(ready[slot])
        texts = []
        for s, c in zip(sides, carries):
            n0 = c.numel()
            buf = s.bufs[slot]
            if n0:
                buf[CARRY_MAX - n0:CARRY_MAX].copy_(c)
            texts.append(buf[CARRY_MAX - n0:CARRY_MAX + s.chunk_len[slot]])
        # every newline of both texts (device); the line counts come back by kernel stores
        for t, cb in zip(texts, cut):
            _lib.check(L.gf_fastq_index_device(h, t.data_ptr(), t.numel(), cb["nl_pos"].data_ptr(), cb["nl_pos"].numel(),
                                               cb["n_lines"].data_ptr(), cb["ws"].data_ptr(), st_main))
            peek.i64(cb["n_lines"], 0)
            peek.i64(cb["n_lines"], 1)
        v = peek.wait()
        lines, newlines = [v[0], v[2]], [v[1], v[3]]
        if max(newlines) > cut[0]["nl_pos"].numel():
            raise _lib.GfError(_lib.GF_ERR_CAPACITY, "a FASTQ chunk of mostly empty lines")
        # A side whose last byte has arrived counts its unterminated last line (fastq_reader.rs:75-147); the
        # others only the lines that end inside the chunk.
        counts = [(ln if f else nl) // 4 for ln, nl, f in zip(lines, newlines, final)]
        m = min(counts)
        for t, cb, nl in zip(texts, cut, newlines):
            if m > 0:
                _lib.check(L.gf_fastq_gather_device(h, t.data_ptr(), t.numel(), cb["nl_pos"].data_ptr(), nl, m,
                                                    cb["offsets"].data_ptr(), cb["bases"].data_ptr(), cb["quals"].data_ptr(),
                                                    cb["bases"].numel(), cb["n_bad"].data_ptr(), cb["ws"].data_ptr(), st_main))
                peek.i64(cb["offsets"], m)
                if 4 * m - 1 < nl:
                    peek.i64(cb["nl_pos"], 4 * m - 1)
        v = peek.wait() if m > 0 else []
        new_carries, used = [], []
        k = 0
        for s, t, nl in zip(sides, texts, newlines):
            if m == 0:
                cut_at, nb = 0, 0
            else:
                nb = v[k]
                k += 1
                if 4 * m - 1 < nl:
                    cut_at = v[k] + 1
                    k += 1
                else:   # record m-1 ends with the text (no final newline)
                    cut_at = t.numel()
            tail = t[cut_at:]
            if tail.numel() > CARRY_MAX:
                raise _lib.GfError(_lib.GF_ERR_CAPACITY, "a FASTQ chunk left more than %d bytes for the next one: the two "
                                   "files' records drift apart faster than the chunks can absorb" % CARRY_MAX)
            new_carries.append(tail.clone())
            s.carry_len = int(tail.numel())
            used.append(nb)
        if m > 0:
            (lc, rc_), (lb, rb_) = cut, used
            args = (lc["bases"][:lb], lc["quals"][:lb], lc["offsets"][:m + 1], rc_["bases"][:rb_], rc_["quals"][:rb_],
                    rc_["offsets"][:m + 1], max_read_len)
            res = scan_pairs_device(indexer, *args, pair_id_base=pairs_done)
            out = _download(res, peek)
            if out[3]["overflow"]:
                res = scan_pairs_device(indexer, *args, pair_id_base=pairs_done, hits_cap=3 * m, bytes_cap=2 * (lb + rb_) + 64,
                                        retry_cap=3 * m)
                out = _download(res, peek)
            out[3]["pairs"] = m
            yield out
        ev = torch.cuda.Event()
        ev.record(main)
        free[slot] = ev
        carries = new_carries
        pairs_done += m
        # records pair up by position and the shorter file ends both (fastq_reader.rs:209-218): stop when a
        # side that has all its bytes has no record left
        if all(final) or any(f and c == m for f, c in zip(final, counts)):
            break
        slot ^= 1


def _download(res, peek: _Peek):
    """PairScan.download without DMA read-backs (falls back to it for lists beyond the pinned block)."""
    for j in range(5):
        peek.i64(res.totals, j)
    t = peek.wait()
    tot = {"hits": t[0], "hit_bytes": t[1], "merged_pairs": t[2], "retried_reads": t[3], "overflow": t[4]}
    k, nb = min(tot["hits"], res.hits.shape[0]), min(tot["hit_bytes"], res.bases.numel())
    raw = peek.bytes(res.hits, 64 * k)
    hb = peek.bytes(res.bases, nb) if raw is not None else None
    hq = peek.bytes(res.quals, nb) if hb is not None else None
    if hq is None:
        return res.download()
    rec = np.frombuffer(raw, dtype=_lib.PAIR_HIT_DTYPE).copy()
    return rec, hb, hq, tot


def scan_pair_end_text(indexer: Indexer, r1_text: np.ndarray, r2_text: np.ndarray, chunk_bytes: int = 128 << 20,
                       threads: int = 8) -> Tuple[List[Tuple[int, ReadMatch]], dict]:
    """The whole paired-end scan of two FASTQ texts up to the ReadMatch list (before the filters):
    ([(pair index, ReadMatch)] in push order, counters)."""
    mapper = FusionMapper(indexer)
    found: List[Tuple[int, ReadMatch]] = []
    counters = {"pairs": 0, "merged_pairs": 0, "retried_reads": 0, "hits": 0, "chunks": 0}
    for rec, hb, hq, tot in scan_pair_text_stream(indexer, r1_text, r2_text, chunk_bytes):
        found += finish_pair_hits(mapper, rec, hb, hq, threads)
        for k in ("pairs", "merged_pairs", "retried_reads", "hits"):
            counters[k] += tot[k]
        counters["chunks"] += 1
    return found, counters
