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

import os
import sys
import threading
import time
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
    free = [None, None]    # per slot: event after which the slot's buffers may be overwritten

    upload_threads = [None, None]

    def start_upload(slot: int):
        """The copy of the slot's next chunk, on a host thread of its own: the calls that queue it block that
        thread, not the one that launches the kernels of the chunk being processed."""
        # (the carry lengths read here are those of the chunk processed LAST, not of the one in flight: the balancing
        #  of the two files lags one chunk behind — harmless, a slot always has room for chunk_bytes behind CARRY_MAX)
        nbytes = [max(chunk_bytes - s.carry_len, 1) for s in sides]
        wait_for = free[slot]

        def run():
            torch.cuda.set_device(dev)
            if wait_for is not None:
                wait_for.synchronize()
            for s, nb in zip(sides, nbytes):   # the file that is ahead (longer carry) gets fewer new bytes
                s.upload(slot, nb, copy_stream)
            copy_stream.synchronize()
        th = threading.Thread(target=run)
        th.start()
        upload_threads[slot] = th

    def wait_upload(slot: int):
        upload_threads[slot].join()

    pairs_done = 0
    slot = 0
    carries = [torch.empty(0, dtype=torch.uint8, device=dev), torch.empty(0, dtype=torch.uint8, device=dev)]
    # The chunks are processed on a stream of their own, not on the legacy null stream: the null stream and
    # the other streams wait for each other, and an upload in flight then stalls every kernel of the chunk
    # being processed (measured: 9.4 ms of upload + 7 ms of processing per 2 x 256 MB, one after the other).
    proc = torch.cuda.Stream(dev)

    def process(slot: int, final, carries, pairs_done: int):
        """One chunk on the stream `proc`: (result or None, new carries, counts, m)."""
        texts = []
        for s, c in zip(sides, carries):
            n0 = c.numel()
            buf = s.bufs[slot]
            if n0:
                buf[CARRY_MAX - n0:CARRY_MAX].copy_(c)
            texts.append(buf[CARRY_MAX - n0:CARRY_MAX + s.chunk_len[slot]])
        # records of both texts (device).  A side whose last byte has arrived counts its unterminated last
        # line (fastq_reader.rs:75-147); the others only the lines that end inside the chunk.
        # (lean: the qualities stay in the chunk's text, which lives in the slot's buffer until the scan below is done)
        batches = [fastq_cut_device(indexer, t, lean=True) for t in texts]
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
        out = None
        if m > 0:
            l, r = batches
            lo, ro = l.offsets[:m + 1], r.offsets[:m + 1]
            lb, rb_ = int(lo[-1].item()), int(ro[-1].item())
            lean = l.qual_off is not None and r.qual_off is not None
            if not lean:   # one side has a quality line of another length than its sequence: both the full way
                l = l if l.qual_off is None else fastq_cut_device(indexer, texts[0])
                r = r if r.qual_off is None else fastq_cut_device(indexer, texts[1])
            lq, rq = (l.quals, r.quals) if lean else (l.quals[:lb], r.quals[:rb_])
            qo = dict(l_qual_off=l.qual_off[:m], r_qual_off=r.qual_off[:m]) if lean else {}
            res = scan_pairs_device(indexer, l.bases[:lb], lq, lo, r.bases[:rb_], rq, ro,
                                    max_read_len, pair_id_base=pairs_done, **qo)
            out = res.download()
            if out[3]["overflow"]:
                res = scan_pairs_device(indexer, l.bases[:lb], lq, lo, r.bases[:rb_], rq, ro,
                                        max_read_len, pair_id_base=pairs_done, hits_cap=3 * m,
                                        bytes_cap=2 * (lb + rb_) + 64, retry_cap=3 * m, **qo)
                out = res.download()
            out[3]["pairs"] = m
        ev = torch.cuda.Event()
        ev.record(proc)
        free[slot] = ev
        return out, new_carries, counts, m

    start_upload(0)
    while True:
        t_a = time.perf_counter()
        wait_upload(slot)                   # this chunk's text is on the device
        t_b = time.perf_counter()
        final = [s.done() for s in sides]   # the side's last byte is in this (or an earlier) chunk
        if not all(final):
            start_upload(slot ^ 1)          # the next chunk crosses the link while this one is processed
        with torch.cuda.stream(proc):
            out, carries, counts, m = process(slot, final, carries, pairs_done)
        if os.environ.get("GF_STREAM_DEBUG") == "1":
            print("chunk: waited %.2f ms for its upload, processed in %.2f ms" % (1e3 * (t_b - t_a), 1e3 * (time.perf_counter() - t_b)),
                  file=sys.stderr, flush=True)
        if out is not None:
            yield out
        pairs_done += m
        # records pair up by position and the shorter file ends both (fastq_reader.rs:209-218): stop when a
        # side that has all its bytes has no record left
        if all(final) or any(f and c == m for f, c in zip(final, counts)):
            break
        slot ^= 1
    for th in upload_threads:
        if th is not None:
            th.join()


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
