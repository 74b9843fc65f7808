"""Multi-CSV mode (BASELINE configs[4]): one resident read set, one index per fusion CSV.

``FusionScan::scan_per_fusion_csv`` (src/core/fusion_scan.rs:62-188) loads the reference and
the FASTQ records once, then scans every CSV of the list as an independent job — its own
``make_index``, its own match list, its own report — on a pool of ``outer`` jobs with
``inner`` threads each: all threads on different CSVs when there are at least as many CSVs as
threads, otherwise ``threads / n_csv`` threads per CSV (:103-110).

The same plan over GPUs: the reads stay resident in HBM, a rank *rebuilds* the index for each
CSV it owns (``gf_index_build``: a few milliseconds) and maps its reads against it.
  * n_csv >= world: CSV k belongs to rank k % world, which maps ALL reads against it — the
    jobs are independent, so there is no collective at all (every rank holds the whole read set);
  * n_csv <  world: ``world // n_csv`` ranks share a CSV, each maps a contiguous shard of the
    reads (dist.shard_range) and the group merges its hit lists with the path's one all-gather.
Per CSV the result is the ordered hit list a single GPU would produce for that CSV.
"""
from __future__ import annotations

from typing import Callable, Dict, List, NamedTuple, Optional, Sequence, Tuple

from .dist import shard_range


class CsvJob(NamedTuple):
    csv: int                 # index into the CSV list
    lo: int                  # this rank maps reads [lo, hi) against it
    hi: int
    group: Tuple[int, ...]   # ranks sharing the CSV, ascending; (rank,) when the rank owns it alone


def plan_multi_csv(n_csv: int, n_reads: int, rank: int, world: int) -> List[CsvJob]:
    """The jobs of ``rank``, in CSV order (fusion_scan.rs:103-110 with ranks for threads)."""
    if n_csv <= 0:
        return []
    if n_csv >= world:
        return [CsvJob(k, 0, n_reads, (rank,)) for k in range(rank, n_csv, world)]
    inner = world // n_csv                     # ranks per CSV; ranks beyond n_csv * inner stay idle,
    k, pos = divmod(rank, inner)               # like the reference's threads beyond outer * inner
    if k >= n_csv:
        return []
    group = tuple(range(k * inner, (k + 1) * inner))
    lo, hi = shard_range(n_reads, pos, inner)
    return [CsvJob(k, lo, hi, group)]


def scan_multi_csv(genesets: Sequence, bases, offsets, max_read_len: int, rank: int = 0, world: int = 1,
                   device: int = -1, groups: Optional[Dict[Tuple[int, ...], object]] = None,
                   on_index: Optional[Callable] = None) -> Dict[int, "object"]:
    """Map the resident reads (device tensors ``bases`` uint8, ``offsets`` int64[n+1]) against
    every gene set this rank owns.  ``genesets[k]`` = (gene slices, reversed flags) of CSV k — what
    ``Indexer.from_gene_slices`` takes.  Returns {csv: hits int64[k, 6]} — for a shared CSV the
    merged list of its group (``groups`` maps a group's rank tuple to its process group).
    ``on_index(csv, indexer)`` is called after each rebuild (tests hook their checks in there)."""
    import torch
    from .dist import allgather_hits
    from .indexer import Indexer
    n = offsets.numel() - 1
    out: Dict[int, object] = {}
    packed = None   # the reads are mapped once per CSV: converted to the kernels' 2-bit form once, by the first index
    for job in plan_multi_csv(len(genesets), n, rank, world):
        seqs, rev = genesets[job.csv]
        ix = Indexer.from_gene_slices(seqs, rev, device=device)
        ix.make_index()                        # index rebuilt per CSV; the reads never move
        try:
            if on_index is not None:
                on_index(job.csv, ix)
            m = job.hi - job.lo
            if packed is None:
                packed = ix.pack_bases_device(bases)
            counts, matches = ix.map_reads_packed_device(packed[0], packed[1], offsets[job.lo:job.hi + 1], max_read_len)
            hits, n_hits = ix.compact_hits_device(counts, matches, m, read_id_base=job.lo, cap=max(m // 8, 4096))
            if len(job.group) > 1:
                merged = allgather_hits(hits, n_hits, group=None if groups is None else groups[job.group])
            else:
                k = int(n_hits.item())
                if k > hits.shape[0]:          # more hits than the first guess: once more with room for all
                    hits, n_hits = ix.compact_hits_device(counts, matches, m, read_id_base=job.lo, cap=k)
                merged = hits[:k]
            out[job.csv] = merged.clone()
            torch.cuda.synchronize()
        finally:
            ix.close()
    return out
