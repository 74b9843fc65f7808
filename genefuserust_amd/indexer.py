"""Host-side mirror of the reference's ``Indexer`` (src/core/indexer.rs:67-608)
over the gfmatch C ABI.

Same names, argument meaning and error behaviour as the Rust type so that the
parity tests read like the reference's own: ``Indexer.with_loaded_ref(reference,
fusions)``, ``make_index()``, ``map_read(read) -> [SeqMatch]``,
``in_required_direction(mapping)``, ``m_fusion_seq``.  On top of that it offers
the batch forms the GPU wants (``map_reads``, ``map_reads_device``).  All
compute goes through libgfmatch.so; there is no Python or CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Mapping, NamedTuple, Optional, Sequence, Tuple, Union

import numpy as np

from . import _lib
from ._lib import GfError, GfHit, GfIndexInfo, GfOptions, GfSeqMatch, HIT_DTYPE, SEQMATCH_DTYPE

BytesLike = Union[bytes, bytearray, str]


class GenePos(NamedTuple):
    """src/core/common.rs:4-7"""
    contig: int
    position: int


class SeqMatch(NamedTuple):
    """src/core/indexer.rs:41-45"""
    seq_start: int
    seq_end: int
    start_gp: GenePos

    def __str__(self) -> str:  # indexer.rs:57-65
        return "%d-%d|%d:%d" % (self.seq_start, self.seq_end, self.start_gp.contig, self.start_gp.position)


@dataclass
class Exon:
    """src/core/gene.rs:9-13"""
    id: int
    start: int
    end: int


@dataclass
class Gene:
    """src/core/gene.rs:16-229: the gene line of a fusion CSV and its exons."""
    m_name: str = "invalid"
    m_chr: str = "invalid"
    m_start: int = 0
    m_end: int = 0
    m_reversed: bool = False
    m_exons: List[Exon] = field(default_factory=list)

    def is_reversed(self) -> bool:
        return self.m_reversed

    def valid(self) -> bool:  # gene.rs:39-41
        return self.m_name != "invalid" and self.m_start != 0 and self.m_end != 0

    @classmethod
    def parse(cls, line_str: str) -> "Gene":
        """gene.rs:43-88: ">NAME,chr:start-end"; anything short of that gives the invalid gene;
        numbers that do not parse raise, like the reference's `?`."""
        splitted = line_str.split(",")
        if len(splitted) < 2:
            return cls()
        name = splitted[0][1:].strip()
        chr_pos = splitted[1].split(":")
        if len(chr_pos) < 2:
            return cls()
        rng = chr_pos[1].split("-")
        if len(rng) < 2:
            return cls()
        return cls(name, chr_pos[0].strip(), _parse_i32(rng[0]), _parse_i32(rng[1]))

    def add_exon(self, id_: int, start: int, end: int) -> None:  # gene.rs:90-105
        self.m_exons.append(Exon(id_, start, end))
        if len(self.m_exons) > 1 and self.m_exons[0].start > self.m_exons[1].start:
            self.m_reversed = True

    def get_exon_intron(self, pos: int) -> Tuple[Optional[bool], Optional[int]]:
        """gene.rs:171-203: (is_exon, number); (None, None) when no exon or intron holds the
        position (the reference leaves its out-parameters untouched)."""
        pp = abs(pos) + self.m_start
        for i, ex in enumerate(self.m_exons):
            if ex.start <= pp <= ex.end:
                return True, ex.id
            if i > 0:
                prev = self.m_exons[0]  # the reference never advances prev_exon: it stays the first exon
                if self.m_reversed:
                    if ex.end < pp < prev.start:
                        return False, ex.id - 1
                elif prev.end < pp < ex.start:
                    return False, ex.id - 1
        return None, None

    def pos2str(self, pos: int) -> str:
        """gene.rs:131-169, e.g. "ALK:exon:20|-chr2:29446222"."""
        pp = abs(pos) + self.m_start
        ss = self.m_name + ":"
        for i, ex in enumerate(self.m_exons):
            if ex.start <= pp <= ex.end:
                ss += "exon:%d|" % ex.id
                break
            if i > 0:
                prev = self.m_exons[i - 1]
                if self.m_reversed:
                    if ex.end < pp < prev.start:
                        ss += "intron:%d|" % (ex.id - 1)
                        break
                elif prev.end < pp < ex.start:
                    ss += "intron:%d|" % (ex.id - 1)
                    break
        return ss + ("+" if pos >= 0 else "-") + "%s:%d" % (self.m_chr, pp)

    def gene_pos_2_chr_pos(self, genepos: int) -> int:  # gene.rs:205-212
        chrpos = abs(genepos) + self.m_start
        return -chrpos if genepos < 0 else chrpos


def _parse_i32(s: str) -> int:
    """Rust's str::parse::<i32> on a trimmed field: optional sign, digits, 32-bit range."""
    t = s.strip()
    body = t[1:] if t[:1] in "+-" else t
    if not body or not body.isascii() or not body.isdigit():
        raise ValueError("invalid digit found in string: %r" % t)
    v = int(t)
    if not -(1 << 31) <= v < (1 << 31):
        raise ValueError("number too large to fit in target type: %r" % t)
    return v


@dataclass
class Fusion:
    """src/core/fusion.rs:12-107"""
    m_gene: Gene

    def is_reversed(self) -> bool:
        return self.m_gene.is_reversed()

    def pos2str(self, pos: int) -> str:
        return self.m_gene.pos2str(pos)

    @staticmethod
    def parse_csv_text(text: str) -> List["Fusion"]:
        """fusion.rs:22-86 on the file's text: a ">NAME,chr:start-end" line opens a gene, the
        "id,start,end" lines after it are its exons; "#" lines, lines with fewer than two fields
        and exon lines with fewer than three are skipped; a gene is kept when it is valid."""
        fusions: List[Fusion] = []
        working = Gene()
        for raw in text.split("\n"):
            line = raw.strip()
            splitted = line.split(",")
            if len(splitted) < 2 or splitted[0].startswith("#"):
                continue
            if splitted[0].startswith(">"):
                if working.valid():
                    fusions.append(Fusion(working))
                working = Gene.parse(line)
                continue
            if len(splitted) < 3:
                continue
            working.add_exon(_parse_i32(splitted[0]), _parse_i32(splitted[1]), _parse_i32(splitted[2]))
        if working.valid():
            fusions.append(Fusion(working))
        return fusions

    @staticmethod
    def parse_csv(filename: str) -> List["Fusion"]:
        with open(filename, "r", encoding="utf-8", newline="") as f:
            return Fusion.parse_csv_text(f.read())


class FastaReader:
    """src/core/fasta_reader.rs:25-200: ``read_all`` fills ``m_all_contigs`` (name -> sequence).
    A record runs from one '>' to the next; its name is the text up to the first newline or
    blank; of the rest only letters, '-' and '*' are kept (newlines and everything else are
    dropped), upper-cased when ``force_upper_case``.  ``.gz`` files are gunzipped."""

    def __init__(self, fasta_file: str, force_upper_case: bool = True):
        import gzip
        import os
        if os.path.isdir(fasta_file):
            raise IsADirectoryError("There is a problem with the provided fasta file: '%s' is a directory NOT a file..."
                                    % fasta_file)
        self.m_fasta_file = str(fasta_file)
        self.m_force_upper_case = force_upper_case
        opener = gzip.open if self.m_fasta_file.endswith(".gz") else open
        with opener(self.m_fasta_file, "rb") as f:
            self._data = f.read()
        if not self._data:
            raise ValueError("empty fasta file: %s" % fasta_file)
        self.m_all_contigs: Dict[str, bytes] = {}

    def read_all(self) -> None:
        keep = bytes(b for b in range(256) if chr(b).isalpha() and b < 128 or b in b"-*")
        drop = bytes(set(range(256)) - set(keep))
        first = self._data.find(b">")          # what precedes the first '>' is skipped (:66-74)
        body = self._data[first + 1:] if first >= 0 else b""
        records = body.split(b">")
        if records and records[-1] == b"":       # nothing after the last '>': read_until returns 0 bytes there
            records.pop()
        for rec in records:                      # read_until(b'>') per record (:125-141)
            cut = len(rec)
            for d in (b"\n", b" "):
                k = rec.find(d)
                if 0 <= k < cut:
                    cut = k
            name = rec[:cut].decode("latin-1")
            seq = rec[cut + 1:].translate(None, drop)   # the delimiter itself is consumed (:144-149)
            if self.m_force_upper_case:
                seq = seq.upper()
            self.m_all_contigs[name] = seq


def _as_bytes(s: BytesLike) -> bytes:
    return s.encode("ascii") if isinstance(s, str) else bytes(s)


def resolve_gene_slice(contigs: Mapping[str, BytesLike], gene: Gene) -> Optional[bytes]:
    """indexer.rs:137-158: chromosome lookup by name, "chr"+name, name without
    "chr"; then the raw CSV numbers used as a half-open byte range.  None when the
    chromosome is missing (the gene then indexes nothing, :149-150).  Like the
    reference's ``.get(a..b).unwrap()`` an out-of-range slice is an error."""
    chr_ = gene.m_chr
    if chr_ not in contigs:
        if "chr" + chr_ in contigs:
            chr_ = "chr" + chr_
        elif chr_.replace("chr", "") in contigs:
            chr_ = chr_.replace("chr", "")
        else:
            return None
    seq = _as_bytes(contigs[chr_])
    if not (0 <= gene.m_start <= gene.m_end <= len(seq)):
        raise IndexError("gene %s: range %d..%d outside contig %s (len %d)"
                         % (gene.m_name, gene.m_start, gene.m_end, chr_, len(seq)))
    return seq[gene.m_start:gene.m_end]


class Indexer:
    """Drop-in for ``struct Indexer``; the index lives in HBM."""

    def __init__(self, reference: Optional[Mapping[str, BytesLike]], fusions: Sequence[Fusion],
                 device: int = -1):
        # Indexer::with_loaded_ref (indexer.rs:100-112)
        self.m_reference = reference
        self.m_fusions = list(fusions)
        self._fusion_seq: Optional[List[str]] = []
        self._device = device
        self._h: Optional[C.c_void_p] = None
        self._gene_slices: Optional[List[Optional[bytes]]] = None

    with_loaded_ref = classmethod(lambda cls, reference, fusions, device=-1: cls(reference, fusions, device))

    @classmethod
    def from_gene_slices(cls, slices: Sequence[Optional[BytesLike]],
                         reversed_flags: Optional[Sequence[bool]] = None, device: int = -1) -> "Indexer":
        """Boundary form: gene slices already cut out (what gf_index_build takes)."""
        fus = [Fusion(Gene("g%d" % i, "", 0, 0, bool(reversed_flags[i]) if reversed_flags else False))
               for i in range(len(slices))]
        ix = cls(None, fus, device)
        ix._gene_slices = [None if s is None else _as_bytes(s) for s in slices]
        return ix

    def get_ref(self):
        return self.m_reference

    # -- make_index (indexer.rs:122-177) ------------------------------------
    def make_index(self) -> None:
        if self._gene_slices is None:
            if self.m_reference is None:
                return  # indexer.rs:123-125
            self._gene_slices = [resolve_gene_slice(self.m_reference, f.m_gene) for f in self.m_fusions]
        L = _lib.lib()
        n = len(self._gene_slices)
        keep = [s if s is not None else b"" for s in self._gene_slices]
        arr = (C.c_char_p * max(n, 1))(*keep) if n else (C.c_char_p * 1)()
        lens = (C.c_int64 * max(n, 1))(*[(-1 if s is None else len(s)) for s in self._gene_slices])
        opts = GfOptions(device=self._device)
        h = C.c_void_p()
        _lib.check(L.gf_index_build(arr, lens, n, C.byref(opts), C.byref(h)))
        self._free()
        self._h = h
        # Fusion::is_reversed() per gene, for the direction rule of the device-resident pair pipeline
        rev = np.array([f.is_reversed() for f in self.m_fusions] or [0], dtype=np.uint8)
        _lib.check(L.gf_index_set_gene_reversed(h, rev.ctypes.data, n))
        self._fusion_seq = None   # fetched from the library on first use (multi-CSV mode never asks)

    @property
    def m_fusion_seq(self) -> List[str]:
        """Indexer.m_fusion_seq (indexer.rs:77, :170): the upper-cased gene slices, "" for an unresolved gene."""
        if self._fusion_seq is None:
            L, h = _lib.lib(), self._handle()
            out = []
            for c in range(len(self._gene_slices or [])):
                ln = L.gf_index_fusion_seq(h, c, None, 0)
                buf = C.create_string_buffer(max(int(ln), 1))
                L.gf_index_fusion_seq(h, c, buf, ln)
                out.append(buf.raw[:ln].decode("latin-1"))
            self._fusion_seq = out
        return self._fusion_seq

    def _handle(self) -> C.c_void_p:
        if self._h is None:
            raise RuntimeError("make_index() has not been called")
        return self._h

    def info(self) -> Dict[str, int]:
        out = GfIndexInfo()
        _lib.check(_lib.lib().gf_index_info_get(self._handle(), C.byref(out)))
        return {k: int(getattr(out, k)) for k, _ in GfIndexInfo._fields_ if k != "pad"}

    def lookup(self, kmers: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """m_kmer_pos / m_dupe_list query for reference-coded k-mers (test aid)."""
        k = np.ascontiguousarray(kmers, dtype=np.uint32)
        n = k.size
        cnt = np.zeros(n, dtype=np.int32)
        ctg = np.zeros((n, 5), dtype=np.int16)
        pos = np.zeros((n, 5), dtype=np.int32)
        _lib.check(_lib.lib().gf_index_lookup(self._handle(), k.ctypes.data, n, cnt.ctypes.data,
                                              ctg.ctypes.data, pos.ctypes.data))
        return cnt, ctg, pos

    # -- map_read (indexer.rs:252-538) --------------------------------------
    def map_read(self, r: BytesLike) -> List[SeqMatch]:
        seq = _as_bytes(getattr(r, "m_seq", r))
        out = (GfSeqMatch * 2)()
        n = _lib.check(_lib.lib().gf_map_read(self._handle(), seq, len(seq), out))
        return [SeqMatch(out[k].seq_start, out[k].seq_end, GenePos(out[k].contig, out[k].position))
                for k in range(n)]

    def map_reads_packed(self, bases: np.ndarray, offsets: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """Batch form on host arrays: returns (counts int32[n], matches SEQMATCH_DTYPE[n,2])."""
        b = np.ascontiguousarray(bases, dtype=np.uint8)
        o = np.ascontiguousarray(offsets, dtype=np.int64)
        n = o.size - 1
        counts = np.zeros(max(n, 0), dtype=np.int32)
        matches = np.zeros((max(n, 0), 2), dtype=SEQMATCH_DTYPE)
        _lib.check(_lib.lib().gf_map_reads(self._handle(), b.ctypes.data if b.size else None, o.ctypes.data, n,
                                           counts.ctypes.data, matches.ctypes.data))
        return counts, matches

    def map_reads(self, reads: Sequence[BytesLike]) -> List[List[SeqMatch]]:
        from .synth import ragged_batch
        bases, offsets = ragged_batch([_as_bytes(r) for r in reads])
        counts, matches = self.map_reads_packed(bases, offsets)
        return unpack_matches(counts, matches)

    def map_reads_hits(self, bases: np.ndarray, offsets: np.ndarray, read_id_base: int = 0,
                       cap: Optional[int] = None) -> np.ndarray:
        """Only the non-empty results, ascending read id (HIT_DTYPE records)."""
        b = np.ascontiguousarray(bases, dtype=np.uint8)
        o = np.ascontiguousarray(offsets, dtype=np.int64)
        n = o.size - 1
        explicit = cap is not None
        cap = cap if explicit else min(n, max(4096, n // 64))  # hits are a fraction of a percent of the reads
        while True:
            hits = np.empty(max(cap, 1), dtype=HIT_DTYPE)
            total = C.c_int64(0)
            _lib.check(_lib.lib().gf_map_reads_hits(self._handle(), b.ctypes.data if b.size else None,
                                                    o.ctypes.data, n, read_id_base, hits.ctypes.data, cap,
                                                    C.byref(total)))
            if explicit or int(total.value) <= cap:
                return hits[:min(int(total.value), cap)]
            cap = int(total.value)  # a batch of mostly hits: once more with room for all of them

    # -- device-resident batch (torch tensors are only carriers of HBM pointers) --
    def map_reads_device(self, bases, offsets, max_read_len: int, counts=None, matches=None, stream=None):
        import torch
        n = offsets.numel() - 1
        assert bases.dtype == torch.uint8 and offsets.dtype == torch.int64 and bases.is_cuda and offsets.is_cuda
        if counts is None:
            counts = torch.empty(max(n, 1), dtype=torch.uint8, device=bases.device)
        if matches is None:
            matches = torch.empty((max(n, 1), 2, 4), dtype=torch.int32, device=bases.device)
        st = torch.cuda.current_stream(bases.device).cuda_stream if stream is None else stream
        _lib.check(_lib.lib().gf_map_reads_device(self._handle(), bases.data_ptr(), offsets.data_ptr(), n,
                                                  int(max_read_len), counts.data_ptr(), matches.data_ptr(), st))
        return counts, matches

    def map_reads_fixed_device(self, bases, read_len: int, counts=None, matches=None, stream=None):
        """A batch of reads of one length laid back to back (gf_map_reads_fixed_device): no offsets array."""
        import torch
        assert bases.dtype == torch.uint8 and bases.is_cuda and bases.numel() % int(read_len) == 0
        n = bases.numel() // int(read_len)
        if counts is None:
            counts = torch.empty(max(n, 1), dtype=torch.uint8, device=bases.device)
        if matches is None:
            matches = torch.empty((max(n, 1), 2, 4), dtype=torch.int32, device=bases.device)
        st = torch.cuda.current_stream(bases.device).cuda_stream if stream is None else stream
        _lib.check(_lib.lib().gf_map_reads_fixed_device(self._handle(), bases.data_ptr(), n, int(read_len), counts.data_ptr(),
                                                        matches.data_ptr(), st))
        return counts, matches

    def pack_bases_device(self, bases, stream=None):
        """The 2-bit + bad-bit form of a whole ``bases`` buffer (gf_pack_bases_device): (pk int32, iv int16)
        tensors; reads are then given by the same offsets.  Worth it when the reads are mapped more than once."""
        import torch
        L = _lib.lib()
        nb = bases.numel()
        ch = int(L.gf_packed_chunks(nb))
        pk = torch.empty(ch, dtype=torch.int32, device=bases.device)
        iv = torch.empty(ch, dtype=torch.int16, device=bases.device)
        st = torch.cuda.current_stream(bases.device).cuda_stream if stream is None else stream
        _lib.check(L.gf_pack_bases_device(self._handle(), bases.data_ptr(), nb, pk.data_ptr(), iv.data_ptr(), st))
        return pk, iv

    def map_reads_packed_device(self, pk, iv, offsets, max_read_len: int, counts=None, matches=None, stream=None):
        """Indexer.map_reads_device on the packed form of the reads (same offsets, same results)."""
        import torch
        n = offsets.numel() - 1
        if counts is None:
            counts = torch.empty(max(n, 1), dtype=torch.uint8, device=pk.device)
        if matches is None:
            matches = torch.empty((max(n, 1), 2, 4), dtype=torch.int32, device=pk.device)
        st = torch.cuda.current_stream(pk.device).cuda_stream if stream is None else stream
        _lib.check(_lib.lib().gf_map_reads_packed_device(self._handle(), pk.data_ptr(), iv.data_ptr(), offsets.data_ptr(), n,
                                                         int(max_read_len), counts.data_ptr(), matches.data_ptr(), st))
        return counts, matches

    def compact_hits_device(self, counts, matches, n: int, read_id_base: int = 0, cap: Optional[int] = None,
                            stream=None, out=None):
        """Ordered compaction on the device; returns (hits int64[cap, 6] view of gf_hit, n_hits tensor).
        ``out`` = (hits, n_hits, workspace) tensors of an earlier call to write into (a steady-state loop
        allocates nothing)."""
        import torch
        dev = counts.device
        cap = n if cap is None else cap
        if out is None:
            hits = torch.empty((max(cap, 1), 6), dtype=torch.int64, device=dev)
            n_hits = torch.zeros(1, dtype=torch.int64, device=dev)
            ws = torch.empty(int(_lib.lib().gf_compact_workspace_bytes(n)), dtype=torch.uint8, device=dev)
        else:
            hits, n_hits, ws = out
            cap = min(cap, hits.shape[0])
            assert ws.numel() >= int(_lib.lib().gf_compact_workspace_bytes(n))
        st = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
        _lib.check(_lib.lib().gf_compact_hits_device(self._handle(), counts.data_ptr(), matches.data_ptr(), n,
                                                     read_id_base, hits.data_ptr(), cap, n_hits.data_ptr(),
                                                     ws.data_ptr(), st))
        if out is None:
            self._last_compact_ws = ws   # (kept alive until the next call: the kernels may still be reading it)
        return hits, n_hits

    # -- in_required_direction (indexer.rs:541-608) ---------------------------
    def in_required_direction(self, mapping: Sequence[SeqMatch]) -> bool:
        n = len(mapping)
        arr = (GfSeqMatch * max(n, 1))()
        for k, m in enumerate(mapping):
            arr[k] = GfSeqMatch(m.seq_start, m.seq_end, m.start_gp.position, m.start_gp.contig, 0)
        rev = np.array([f.is_reversed() for f in self.m_fusions] or [0], dtype=np.uint8)
        return bool(_lib.check(_lib.lib().gf_in_required_direction(arr, n, rev.ctypes.data, len(self.m_fusions))))

    def set_profiling(self, on: bool) -> None:
        _lib.check(_lib.lib().gf_set_profiling(self._handle(), int(on)))

    def set_map_variant(self, variant: int) -> None:
        """0 = flat pipeline (default), 1 = wave-per-read probe-all, 2 = wave-per-read seed+verify."""
        _lib.check(_lib.lib().gf_set_map_variant(self._handle(), int(variant)))

    def set_pack_call_reads(self, reads: int) -> None:
        """Host-buffer calls of up to `reads` reads take the zero-copy route (gfmatch.h); -1 = default, 0 = batch route."""
        _lib.check(_lib.lib().gf_set_pack_call_reads(self._handle(), int(reads)))

    def last_stage_ms(self):
        """Flat pipeline: ms of its four kernels (seed+verify, filter, buckets, exact kernel)."""
        import ctypes as C
        out = (C.c_float * 4)()
        _lib.check(_lib.lib().gf_last_stage_ms(self._handle(), out))
        return [float(x) for x in out]

    def last_map_kernel_ms(self) -> float:
        return float(_lib.lib().gf_last_map_kernel_ms(self._handle()))

    def _free(self) -> None:
        if self._h is not None:
            _lib.lib().gf_index_free(self._h)
            self._h = None

    def close(self) -> None:
        self._free()

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass


def unpack_matches(counts: np.ndarray, matches: np.ndarray) -> List[List[SeqMatch]]:
    out: List[List[SeqMatch]] = []
    for r in range(counts.size):
        row = []
        for k in range(int(counts[r])):
            m = matches[r, k]
            row.append(SeqMatch(int(m["seq_start"]), int(m["seq_end"]),
                                GenePos(int(m["contig"]), int(m["position"]))))
        out.append(row)
    return out


def hits_to_numpy(hits_i64) -> np.ndarray:
    """torch int64[k, 6] view of gf_hit records -> structured numpy array."""
    a = hits_i64.detach().cpu().numpy()
    return a.view(HIT_DTYPE).reshape(-1)
