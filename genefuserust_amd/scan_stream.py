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

    def __init__(self, text: np.ndarray, chunk_bytes: int, dev, handle):
        import torch
        self.h = handle
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
            # through the library's own hipMemcpyAsync: torch only treats memory of its own pinned allocator as
            # pinned, and copies from anything else (gf_host_alloc memory included) synchronously
            src = self.text[self.pos:self.pos + n]
            _lib.check(_lib.lib().gf_copy_from_host_device(self.h, src.ctypes.data, self.bufs[slot].data_ptr() + CARRY_MAX, n,
                                                           stream.cuda_stream))
        self.pos += n


def scan_pair_text_stream(indexer: Indexer, r1_text: np.ndarray, r2_text: np.ndarray, chunk_bytes: int = 128 << 20,
                          max_read_len: int = 320) -> Iterator[Tuple[np.ndarray, bytes, bytes, dict]]:
    """Yields, per chunk, what ``PairScan.download`` returns — (gf_pair_hit records with pair ids counted
    from the start of the files, the matched reads' bases, their qualities, totals) — for the FASTQ
    texts ``r1_text`` / ``r2_text`` (uint8 arrays; pinned memory makes the copies asynchronous).
    Records pair up by position; the shorter file ends both (fastq_reader.rs:209-218)."""
    import torch
    from .fastq import fastq_cut_device
    dev = torch.device("cuda", indexer.info()["device"])
    copy_stream = torch.cuda.Stream(dev)
    main = torch.cuda.current_stream(dev)
    sides = [_Side(r1_text, chunk_bytes, dev, indexer._handle()), _Side(r2_text, chunk_bytes, dev, indexer._handle())]
    L, h = _lib.lib(), indexer._handle()
    ready = [None, None]   # per slot: event after which the slot's copies have landed
    free = [None, None]    # per slot: event after which the slot's buffers may be overwritten

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
    final = [False, False]   # the side's last byte is in the current (or an earlier) chunk
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
        # records of both texts (device).  A side whose last byte has arrived counts its unterminated last
        # line (fastq_reader.rs:75-147); the others only the lines that end inside the chunk.
        batches = [fastq_cut_device(indexer, t) for t in texts]
        counts = [b.n_records if f else b.n_newlines // 4 for b, f in zip(batches, final)]
        m = min(counts)
        new_carries = []
        for s, b, t in zip(sides, batches, texts):
            if m == 0:
                cut = 0
            elif 4 * m - 1 < b.n_newlines:
                cut = int(b.nl_pos[4 * m - 1].item()) + 1
            else:   # record m-1 ends with the text (no final newline)
                cut = t.numel()
            tail = t[cut:]
            if tail.numel() > CARRY_MAX:
                raise _lib.GfError(_lib.GF_ERR_CAPACITY, "a FASTQ chunk left more than %d bytes for the next one: the two "
                                   "files' records drift apart faster than the chunks can absorb" % CARRY_MAX)
            new_carries.append(tail.clone())
            s.carry_len = int(tail.numel())
        if m > 0:
            l, r = batches
            lo, ro = l.offsets[:m + 1], r.offsets[:m + 1]
            lb, rb_ = int(lo[-1].item()), int(ro[-1].item())
            res = scan_pairs_device(indexer, l.bases[:lb], l.quals[:lb], lo, r.bases[:rb_], r.quals[:rb_], ro,
                                    max_read_len, pair_id_base=pairs_done)
            out = res.download()
            if out[3]["overflow"]:
                res = scan_pairs_device(indexer, l.bases[:lb], l.quals[:lb], lo, r.bases[:rb_], r.quals[:rb_], ro,
                                        max_read_len, pair_id_base=pairs_done, hits_cap=3 * m,
                                        bytes_cap=2 * (lb + rb_) + 64, retry_cap=3 * m)
                out = res.download()
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
