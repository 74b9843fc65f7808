"""SURVEY.md §8(f)-2 — FASTQ ingest: ``FastqReader`` / ``FastqReaderPair``
(src/core/fastq_reader.rs:29-219) with the record cutting done on the device.

The host reads the file (gunzips it when the name says so — the same extension rule as
``is_zip_fastq`` / ``is_fastq``, fastq_reader.rs:149-179), ships the plain text to HBM
once, and ``gf_fastq_index_device`` + ``gf_fastq_gather_device`` (csrc/gf_fastq_kernels.h)
turn it into the ``bases`` / ``quals`` / ``offsets`` layout the merge and mapping kernels
take.  No CPU fallback: without the HIP library and a GPU every call here raises.
"""
from __future__ import annotations

import gzip
from typing import List, NamedTuple, Optional, Tuple

import numpy as np

from . import _lib
from .indexer import Indexer

ZIP_EXT = (".fastq.gz", ".fq.gz", ".fasta.gz", ".fa.gz")   # fastq_reader.rs:149-163
PLAIN_EXT = (".fastq", ".fq", ".fasta", ".fa")             # fastq_reader.rs:165-179


class FastqBatch(NamedTuple):
    """Records of one FASTQ text, device resident.  ``bases``/``quals`` uint8, ``offsets``
    int64[n+1]; ``nl_pos`` int64[newlines] locates every line (record i = lines 4i..4i+3)."""
    bases: "object"
    quals: "object"
    offsets: "object"
    n_records: int
    nl_pos: "object"
    n_newlines: int
    n_bad_quality: int
    qual_off: "object" = None   # lean cut: quals is the text, and this is where each record's quality line starts

    def max_read_len(self) -> int:
        if self.n_records == 0:
            return 0
        return int((self.offsets[1:] - self.offsets[:-1]).max().item())


def fastq_cut_device(indexer: Indexer, text, stream=None, lean: bool = False) -> FastqBatch:
    """``FastqReader::read`` until it returns None, for a text already in HBM (uint8 tensor).

    ``lean``: the qualities stay in the text (gf_fastq_gather_lean_device) — ``quals`` is then the text itself and
    ``qual_off`` (int64[n_records]) says where each record's quality line starts; what scan_pairs_device takes as
    ``l_qual_off`` / ``r_qual_off``.  A text with a quality line of another length than its sequence (the reference
    does not check; n_bad_quality counts them) is cut the full way instead, and ``qual_off`` is None."""
    import torch
    assert text.dtype == torch.uint8 and text.is_cuda and text.dim() == 1
    dev = text.device
    n = text.numel()
    L, h = _lib.lib(), indexer._handle()
    st = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
    ws = torch.empty(int(L.gf_fastq_workspace_bytes(n)), dtype=torch.uint8, device=dev)
    n_lines = torch.zeros(2, dtype=torch.int64, device=dev)
    cap = n // 16 + 1024   # FASTQ lines average far more than 16 bytes; grown below if not
    nl_pos = torch.empty(cap, dtype=torch.int64, device=dev)
    _lib.check(L.gf_fastq_index_device(h, text.data_ptr(), n, nl_pos.data_ptr(), cap, n_lines.data_ptr(),
                                       ws.data_ptr(), st))
    lines, newlines = (int(x) for x in n_lines.cpu())
    if newlines > cap:   # a text of mostly empty lines
        cap = newlines
        nl_pos = torch.empty(cap, dtype=torch.int64, device=dev)
        _lib.check(L.gf_fastq_index_device(h, text.data_ptr(), n, nl_pos.data_ptr(), cap, n_lines.data_ptr(),
                                           ws.data_ptr(), st))
    n_rec = lines // 4
    offsets = torch.zeros(n_rec + 1, dtype=torch.int64, device=dev)
    bases = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)
    n_bad = torch.zeros(1, dtype=torch.int64, device=dev)
    if lean:
        qual_off = torch.empty(max(n_rec, 1), dtype=torch.int64, device=dev)
        _lib.check(L.gf_fastq_gather_lean_device(h, text.data_ptr(), n, nl_pos.data_ptr(), newlines, n_rec,
                                                 offsets.data_ptr(), bases.data_ptr(), n, qual_off.data_ptr(),
                                                 n_bad.data_ptr(), ws.data_ptr(), st))
        total, bad = int(offsets[-1].item()), int(n_bad.item())
        if bad == 0:
            return FastqBatch(bases[:total], text, offsets, n_rec, nl_pos[:newlines], newlines, 0, qual_off[:n_rec])
    quals = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)
    _lib.check(L.gf_fastq_gather_device(h, text.data_ptr(), n, nl_pos.data_ptr(), newlines, n_rec, offsets.data_ptr(),
                                        bases.data_ptr(), quals.data_ptr(), n, n_bad.data_ptr(), ws.data_ptr(), st))
    total = int(offsets[-1].item())
    return FastqBatch(bases[:total], quals[:total], offsets, n_rec, nl_pos[:newlines], newlines, int(n_bad.item()))


class FastqReader:
    """fastq_reader.rs:19-184.  ``read_all_device`` replaces the ``read()`` loop."""

    def __init__(self, file_name: str, has_quality: bool = True):
        self.m_filename = str(file_name)
        self.m_has_quality = has_quality
        if self.m_filename.endswith(ZIP_EXT):
            self.m_zipped = True
        elif self.m_filename.endswith(PLAIN_EXT):
            self.m_zipped = False
        else:  # the reference prints this and exits (fastq_reader.rs:54-55)
            raise ValueError("ERROR: the input file should be fastq (.fq, .fastq) or gzipped fastq (.fq.gz, .fastq.gz) "
                             + self.m_filename)
        if not has_quality:
            raise NotImplementedError("three-line records (has_quality = false) are not used by the scanners")

    def text(self) -> bytes:
        if self.m_zipped:
            with gzip.open(self.m_filename, "rb") as f:   # MultiGzDecoder: concatenated members too
                return f.read()
        with open(self.m_filename, "rb") as f:
            return f.read()

    def read_all_device(self, indexer: Indexer) -> Tuple[FastqBatch, bytes]:
        """(batch, host text).  The host text is kept for names and strand lines."""
        import torch
        t = self.text()
        dev = torch.device("cuda", indexer.info()["device"])
        d = torch.from_numpy(np.frombuffer(t, dtype=np.uint8).copy()).to(dev) if t else \
            torch.empty(0, dtype=torch.uint8, device=dev)
        return fastq_cut_device(indexer, d), t


def record_lines(batch: FastqBatch, text: bytes, i: int) -> Tuple[bytes, bytes, bytes, bytes]:
    """(name, sequence, strand, quality) of record i, cut from the host text by the device's
    newline index."""
    lo, hi = max(4 * i - 1, 0), min(4 * i + 4, batch.n_newlines)
    nl = batch.nl_pos[lo:hi].cpu().numpy()
    out: List[bytes] = []
    for line in range(4 * i, 4 * i + 4):
        start = 0 if line == 0 else int(nl[line - 1 - lo]) + 1
        end = int(nl[line - lo]) if line < batch.n_newlines else len(text)
        out.append(text[start:end])
    return tuple(out)  # type: ignore[return-value]


class FastqReaderPair:
    """fastq_reader.rs:186-219: records are paired by position; the shorter file ends both."""

    def __init__(self, left: FastqReader, right: FastqReader):
        self.m_left, self.m_right = left, right

    @classmethod
    def from_paths(cls, left_name: str, right_name: str) -> "FastqReaderPair":
        return cls(FastqReader(left_name, True), FastqReader(right_name, True))

    def read_all_device(self, indexer: Indexer):
        (l, lt), (r, rt) = self.m_left.read_all_device(indexer), self.m_right.read_all_device(indexer)
        n = min(l.n_records, r.n_records)
        if l.n_records != n:
            l = l._replace(offsets=l.offsets[:n + 1], n_records=n, bases=l.bases[:int(l.offsets[n].item())],
                           quals=l.quals[:int(l.offsets[n].item())])
        if r.n_records != n:
            r = r._replace(offsets=r.offsets[:n + 1], n_records=n, bases=r.bases[:int(r.offsets[n].item())],
                           quals=r.quals[:int(r.offsets[n].item())])
        return (l, lt), (r, rt)
