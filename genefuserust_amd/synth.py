"""Seeded synthetic inputs for the Indexer hot path.

The reference's real inputs (hg38.fa, druggable/cancer.hg38.csv, the benchmark
FASTQ) are git-ignored upstream, so benchmarks and large parity tests use
index *shapes* taken from the reference's test CSVs (data/index_shapes.json, made
by tools/make_index_shapes.py) filled with seeded random sequence, and read
mixes drawn from those genes.  Everything here is tensor code (torch) so the
same generator runs on the CPU for tests and on the GPU for 10M-pair batches.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np
import torch

_SHAPES = None
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def index_shapes() -> dict:
    global _SHAPES
    if _SHAPES is None:
        here = os.path.dirname(os.path.abspath(__file__))
        with open(os.path.join(here, "data", "index_shapes.json")) as f:
            _SHAPES = json.load(f)
    return _SHAPES


@dataclass
class GeneSet:
    """Gene slices as the reference's make_index sees them (indexer.rs:154-159)."""
    names: List[str]
    seqs: List[bytes]  # raw gene slices (may contain N; upper case)
    reversed_flags: List[bool]

    @property
    def total_bp(self) -> int:
        return sum(len(s) for s in self.seqs)


_LOW_COMPLEXITY_UNITS = (b"A", b"T", b"CA", b"GT", b"GATA", b"AAAG", b"CCG", b"TTAGGG")


def make_gene(length: int, seed: int, family: Optional[np.ndarray] = None,
              repeat_frac: float = 0.02, n_frac: float = 0.001, low_complexity_frac: float = 0.0) -> bytes:
    """i.i.d. uniform ACGT, then `repeat_frac` of the gene overwritten by copies of
    300-bp elements of a shared 20-element family (gives 2..5-fold and >=6-fold
    k-mers), then `low_complexity_frac` of it by poly-A / tandem-repeat stretches of
    60..400 bases (homopolymers, di-, tri-, tetra- and hexanucleotide units: every window
    inside one is a >= 6-fold key, and their edges give short-period near-repeats), then
    `n_frac` of the bases set to N."""
    rng = np.random.default_rng(seed)
    g = _ACGT[rng.integers(0, 4, size=length, dtype=np.int64)].copy()
    if family is not None and length > 600:
        n_copies = int(length * repeat_frac / family.shape[1])
        for _ in range(n_copies):
            e = family[rng.integers(0, family.shape[0])]
            p = int(rng.integers(0, length - family.shape[1]))
            g[p:p + family.shape[1]] = e
    if low_complexity_frac > 0 and length > 1000:
        covered = 0
        while covered < length * low_complexity_frac:
            unit = np.frombuffer(_LOW_COMPLEXITY_UNITS[int(rng.integers(0, len(_LOW_COMPLEXITY_UNITS)))], dtype=np.uint8)
            ln = int(rng.integers(60, 401))
            p = int(rng.integers(0, length - ln))
            g[p:p + ln] = np.resize(unit, ln)
            covered += ln
    n_n = int(length * n_frac)
    if n_n:
        g[rng.integers(0, length, size=n_n)] = ord("N")
    return g.tobytes()


def make_geneset(shape: str = "IDX-D", scale: float = 1.0, seed: int = 1000, repeat_frac: float = 0.02,
                 low_complexity_frac: float = 0.0) -> GeneSet:
    """`shape` in {IDX-T, IDX-D, IDX-C}; `scale` < 1 shrinks every gene (tests); `repeat_frac` /
    `low_complexity_frac`: see make_gene (stress of the rare path: more reads survive to the exact kernel)."""
    genes = index_shapes()[shape]
    fam_rng = np.random.default_rng(seed - 1)
    family = _ACGT[fam_rng.integers(0, 4, size=(20, 300))]
    names, seqs, rev = [], [], []
    for gi, g in enumerate(genes):
        ln = max(64, int(g["len"] * scale))
        names.append(g["name"])
        seqs.append(make_gene(ln, seed + gi, family, repeat_frac=repeat_frac, low_complexity_frac=low_complexity_frac))
        rev.append(bool(g["reversed"]))
    return GeneSet(names, seqs, rev)


_COMP_LUT = np.full(256, ord("N"), dtype=np.uint8)
for _a, _b in zip(b"ACGTacgt", b"TGCATGCA"):
    _COMP_LUT[_a] = _b


@dataclass
class ReadBatch:
    bases: torch.Tensor    # uint8 [total], concatenated ASCII
    offsets: torch.Tensor  # int64 [n+1]
    kinds: torch.Tensor    # uint8 [n]: 0 background, 1 single gene, 2 junction

    @property
    def n(self) -> int:
        return self.offsets.numel() - 1


MIXES = {
    # background, single-gene, junction
    "PANEL": (0.40, 0.599, 0.001),
    "WGS": (0.995, 0.0049, 0.0001),
}


def make_reads(genes: GeneSet, n: int, read_len: int = 150, mix: str = "PANEL",
               seed: int = 20240116, device: str = "cpu", chunk: int = 1 << 20) -> ReadBatch:
    """Fixed-length reads drawn per §8(d): background = uniform random ACGT;
    single-gene = uniform gene/strand/offset with 0.5 % substitutions, 0.2 % of them
    holding one N; junction = two genes joined at a break in [30, L-30], 10 % with a
    1-bp deletion 20 bp left of the break; every read reverse-complemented with
    probability 1/2 (R2-like mates)."""
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    L = read_len
    cat = np.frombuffer(b"".join(genes.seqs), dtype=np.uint8)
    G = torch.from_numpy(cat.copy()).to(dev)
    glen = torch.tensor([len(s) for s in genes.seqs], dtype=torch.int64, device=dev)
    goff = torch.cumsum(glen, 0) - glen
    usable = (glen >= 2 * L).nonzero().flatten()
    if usable.numel() == 0:
        raise ValueError("no gene long enough for reads of length %d" % L)
    acgt = torch.from_numpy(_ACGT.copy()).to(dev)
    comp = torch.from_numpy(_COMP_LUT.copy()).to(dev)
    p_bg, p_single, p_junc = MIXES[mix]
    out = torch.empty((n, L), dtype=torch.uint8, device=dev)
    kinds = torch.empty(n, dtype=torch.uint8, device=dev)
    ar = torch.arange(L, device=dev, dtype=torch.int64)

    def rnd(shape):
        return torch.rand(shape, generator=gen, device=dev)

    def rint(hi, shape):
        return torch.randint(0, hi, shape, generator=gen, device=dev, dtype=torch.int64)

    for c0 in range(0, n, chunk):
        m = min(chunk, n - c0)
        u = rnd((m,))
        kind = torch.where(u < p_bg, 0, torch.where(u < p_bg + p_single, 1, 2)).to(torch.uint8)
        # background everywhere first
        reads = acgt[rint(4, (m, L))]
        # single-gene reads
        g1 = usable[rint(usable.numel(), (m,))]
        o1 = (rnd((m,)) * (glen[g1] - L).to(torch.float64)).to(torch.int64).clamp_(min=0)
        src1 = (goff[g1] + o1)[:, None] + ar[None, :]
        single = G[src1]
        sub = rnd((m, L)) < 0.005
        single = torch.where(sub, acgt[rint(4, (m, L))], single)
        has_n = rnd((m,)) < 0.002
        npos = rint(L, (m,))
        single = torch.where(has_n[:, None] & (ar[None, :] == npos[:, None]),
                             torch.full_like(single, ord("N")), single)
        # junction reads: left = gene A ending at pa (b bases), right = gene B from pb
        g2 = usable[rint(usable.numel(), (m,))]
        brk = 30 + rint(L - 60 + 1, (m,))  # bases taken from the left gene
        pa = L + (rnd((m,)) * (glen[g1] - 2 * L).to(torch.float64)).to(torch.int64).clamp_(min=0)
        pb = (rnd((m,)) * (glen[g2] - L).to(torch.float64)).to(torch.int64).clamp_(min=0)
        dele = rnd((m,)) < 0.10
        left_idx = pa[:, None] - (brk[:, None] - 1 - ar[None, :])  # read pos j <- A[pa-(b-1-j)]
        # 1-bp deletion 20 bp left of the break: bases before it come from one further left
        left_idx = torch.where(dele[:, None] & (ar[None, :] < (brk - 20)[:, None]), left_idx - 1, left_idx)
        right_idx = pb[:, None] + (ar[None, :] - brk[:, None])
        is_left = ar[None, :] < brk[:, None]
        src2 = torch.where(is_left, goff[g1][:, None] + left_idx, goff[g2][:, None] + right_idx)
        src2 = src2.clamp_(0, G.numel() - 1)
        junction = G[src2]
        reads = torch.where((kind == 1)[:, None], single, reads)
        reads = torch.where((kind == 2)[:, None], junction, reads)
        # mate-like orientation: half of all reads reverse-complemented
        flip = rnd((m,)) < 0.5
        rc = comp[reads.to(torch.int64)].flip(1)
        reads = torch.where(flip[:, None], rc, reads)
        out[c0:c0 + m] = reads
        kinds[c0:c0 + m] = kind
    offsets = torch.arange(n + 1, device=dev, dtype=torch.int64) * L
    return ReadBatch(out.reshape(-1), offsets, kinds)


def ragged_batch(seqs: List[bytes]) -> Tuple[np.ndarray, np.ndarray]:
    """Concatenate arbitrary reads into (bases uint8[total], offsets int64[n+1])."""
    lens = np.array([len(s) for s in seqs], dtype=np.int64)
    offsets = np.zeros(len(seqs) + 1, dtype=np.int64)
    np.cumsum(lens, out=offsets[1:])
    bases = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy()
    return bases, offsets


@dataclass
class PairBatch:
    """Read pairs as the FASTQ cut leaves them: R1 = l_*, R2 = r_* (not reverse-complemented),
    fixed read length, so both files share one offsets array."""
    l_bases: torch.Tensor
    l_quals: torch.Tensor
    r_bases: torch.Tensor
    r_quals: torch.Tensor
    offsets: torch.Tensor
    kinds: torch.Tensor      # 0 background, 1 single gene, 2 junction
    frag_len: torch.Tensor

    @property
    def n(self) -> int:
        return self.offsets.numel() - 1


def make_pairs(genes: GeneSet, n: int, read_len: int = 150, mix: str = "PANEL", seed: int = 20240116,
               device: str = "cpu", chunk: int = 1 << 19, frag_mean: float = 300.0, frag_sd: float = 30.0) -> PairBatch:
    """Paired reads per SURVEY.md §8(d): a fragment of N(300, 30) bases clipped to [read_len, 500] —
    background (random), from one gene, or across a junction of two genes (break at least 40 bases from
    either end) — on either strand; R1 = its first read_len bases, R2 = the reverse complement of its
    last read_len bases.  0.5 % of the bases of each read are substituted and carry quality '#', the rest
    'F', so that overlapping mates still merge (fast_merge accepts a mismatch of a high against a low
    quality, read.rs:313-440)."""
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    L, FMAX = read_len, 500
    cat = np.frombuffer(b"".join(genes.seqs), dtype=np.uint8)
    G = torch.from_numpy(cat.copy()).to(dev)
    glen = torch.tensor([len(s) for s in genes.seqs], dtype=torch.int64, device=dev)
    goff = torch.cumsum(glen, 0) - glen
    usable = (glen >= 2 * FMAX).nonzero().flatten()
    if usable.numel() == 0:
        raise ValueError("no gene long enough for fragments of %d bases" % FMAX)
    acgt = torch.from_numpy(_ACGT.copy()).to(dev)
    comp = torch.from_numpy(_COMP_LUT.copy()).to(dev)
    p_bg, p_single, _ = MIXES[mix]
    ar = torch.arange(FMAX, device=dev, dtype=torch.int64)
    arL = torch.arange(L, device=dev, dtype=torch.int64)
    out = {k: torch.empty((n, L), dtype=torch.uint8, device=dev) for k in ("lb", "lq", "rb", "rq")}
    kinds = torch.empty(n, dtype=torch.uint8, device=dev)
    flens = torch.empty(n, dtype=torch.int64, device=dev)

    def rnd(shape):
        return torch.rand(shape, generator=gen, device=dev)

    def rint(hi, shape):
        return torch.randint(0, hi, shape, generator=gen, device=dev, dtype=torch.int64)

    for c0 in range(0, n, chunk):
        m = min(chunk, n - c0)
        u = rnd((m,))
        kind = torch.where(u < p_bg, 0, torch.where(u < p_bg + p_single, 1, 2)).to(torch.uint8)
        flen = (frag_mean + frag_sd * torch.randn((m,), generator=gen, device=dev)).round().to(torch.int64).clamp_(L, FMAX)
        frag = acgt[rint(4, (m, FMAX))]
        g1 = usable[rint(usable.numel(), (m,))]
        g2 = usable[rint(usable.numel(), (m,))]
        o1 = (rnd((m,)) * (glen[g1] - FMAX).to(torch.float64)).to(torch.int64).clamp_(min=0)
        o2 = (rnd((m,)) * (glen[g2] - FMAX).to(torch.float64)).to(torch.int64).clamp_(min=0)
        single = G[(goff[g1] + o1)[:, None] + ar[None, :]]
        brk = 40 + (rnd((m,)) * (flen - 80).to(torch.float64)).to(torch.int64)
        right = G[((goff[g2] + o2)[:, None] + (ar[None, :] - brk[:, None])).clamp_(0, G.numel() - 1)]
        junction = torch.where(ar[None, :] < brk[:, None], single, right)
        frag = torch.where((kind == 1)[:, None], single, frag)
        frag = torch.where((kind == 2)[:, None], junction, frag)
        flip = rnd((m,)) < 0.5   # the fragment's other strand
        rc_frag = comp[torch.gather(frag, 1, (flen[:, None] - 1 - ar[None, :]).clamp_(min=0)).to(torch.int64)]
        frag = torch.where(flip[:, None], rc_frag, frag)
        r1 = frag[:, :L]
        r2 = comp[torch.gather(frag, 1, flen[:, None] - 1 - arL[None, :]).to(torch.int64)]
        for name, rd in (("l", r1), ("r", r2)):
            err = rnd((m, L)) < 0.005
            sub = acgt[rint(4, (m, L))]
            out[name + "b"][c0:c0 + m] = torch.where(err, sub, rd)
            out[name + "q"][c0:c0 + m] = torch.where(err, torch.full_like(rd, ord("#")), torch.full_like(rd, ord("F")))
        kinds[c0:c0 + m] = kind
        flens[c0:c0 + m] = flen
    offsets = torch.arange(n + 1, device=dev, dtype=torch.int64) * L
    return PairBatch(out["lb"].reshape(-1), out["lq"].reshape(-1), out["rb"].reshape(-1), out["rq"].reshape(-1), offsets,
                     kinds, flens)


def make_pair_reads(genes: GeneSet, n_pairs: int, read_len: int = 150, mix: str = "PANEL", seed: int = 20240116,
                    device: str = "cpu", block: int = 1 << 21, n_frac: float = 0.002) -> ReadBatch:
    """The reads of `make_pairs` as ONE batch in FASTQ order — read 2i = R1 of pair i, read 2i+1 = its R2
    (the reverse complement of the fragment's far end) — which is how a pair-end scan presents them to
    Indexer::map_read (pescanner.rs:473-515).  SURVEY.md §8(d): fragments N(300, 30); 0.2 % of the reads from
    a gene hold one N.  Made block by block (own seed each) so that 100 M pairs never hold their qualities."""
    dev = torch.device(device)
    L = read_len
    out = torch.empty((n_pairs, 2, L), dtype=torch.uint8, device=dev)
    kinds = torch.empty((n_pairs, 2), dtype=torch.uint8, device=dev)
    gen = torch.Generator(device=dev)
    for k, p0 in enumerate(range(0, n_pairs, block)):
        m = min(block, n_pairs - p0)
        pb = make_pairs(genes, m, read_len=L, mix=mix, seed=seed + 7919 * k, device=device)
        out[p0:p0 + m, 0] = pb.l_bases.view(m, L)
        out[p0:p0 + m, 1] = pb.r_bases.view(m, L)
        kinds[p0:p0 + m] = pb.kinds[:, None]
        del pb
        if n_frac > 0:
            gen.manual_seed(seed + 7919 * k + 1)
            blk = out[p0:p0 + m].view(2 * m, L)
            has_n = (torch.rand(2 * m, generator=gen, device=dev) < n_frac) & (kinds[p0:p0 + m].reshape(-1) > 0)
            rows = has_n.nonzero().flatten()
            cols = torch.randint(0, L, (rows.numel(),), generator=gen, device=dev)
            blk[rows, cols] = ord("N")
    offsets = torch.arange(2 * n_pairs + 1, device=dev, dtype=torch.int64) * L
    return ReadBatch(out.reshape(-1), offsets, kinds.reshape(-1))
