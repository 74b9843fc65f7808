"""Multi-GPU form of the hot path: reads shard, the index replicates, one exchange.

Reads are independent and the index is read-only (Indexer::map_read takes
``&self``; the reference shares it across consumer threads,
pescanner.rs:296-311), so a batch of n reads is cut into contiguous shards, one
per rank (one process per GPU).  Every rank builds the same index from the same
gene slices (identical lookup results by construction; no broadcast needed) and
maps its shard with no data-path collective.  The single exchange step is the
merge of the per-rank hit lists before host-side scoring: an all-gather of the
hit counts (8 B per rank) followed by one all-gather of the hit records padded
to the largest count (RCCL over xGMI when the backend is "nccl"; KBs to a few
MBs, latency-bound).  Shards are contiguous and hit lists are ordered, so
concatenating the valid prefixes in rank order gives the global read order —
the same list a single GPU produces.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist

HIT_WORDS = 6  # gf_hit is 48 bytes = 6 int64 words; word 0 = read_id


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of rank `rank`."""
    return (rank * n) // world, ((rank + 1) * n) // world


def allgather_hits(hits: torch.Tensor, n_hits: torch.Tensor, group=None) -> torch.Tensor:
    """Merge per-rank ordered hit lists into the global ordered list on every rank.

    hits   int64[cap, 6] (gf_hit records; only the first n_hits rows are valid)
    n_hits int64[1] on the same device
    """
    world = dist.get_world_size(group)
    if world == 1:
        return hits[: int(n_hits.item())]
    if hits.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal on a box without RCCL peers: stage through the host
        return allgather_hits(hits.cpu(), n_hits.cpu(), group).to(hits.device)
    counts = torch.empty(world, dtype=torch.int64, device=hits.device)
    dist.all_gather_into_tensor(counts, n_hits.reshape(1), group=group)
    counts_h = counts.cpu()
    mx = int(counts_h.max().item())
    if mx == 0:
        return hits[:0]
    mine = int(counts_h[dist.get_rank(group)])
    if hits.shape[0] < mine:
        raise ValueError("hit buffer smaller than its own count")
    if hits.shape[0] >= mx:
        send = hits[:mx].contiguous()
    else:  # this rank's buffer is shorter than the largest list: pad to the common length
        send = torch.zeros((mx, HIT_WORDS), dtype=torch.int64, device=hits.device)
        send[:mine] = hits[:mine]
    recv = torch.empty((world, mx, HIT_WORDS), dtype=torch.int64, device=hits.device)
    dist.all_gather_into_tensor(recv.view(world * mx, HIT_WORDS), send, group=group)
    parts = [recv[r, : int(counts_h[r])] for r in range(world)]
    return torch.cat(parts, dim=0)
