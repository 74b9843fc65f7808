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


class HitExchange:
    """The same merge without a host round trip, so that it overlaps the next batch.

    ``allgather_hits`` reads the per-rank counts on the host before it can size the second
    collective — one synchronisation per batch, which at ~3 ms of kernels per 20 M reads costs
    a tenth of the step.  Here every rank sends a fixed-capacity block (row 0 = its count, then
    up to ``cap`` records) in ONE asynchronous all-gather; the valid prefixes are packed into
    rank order on the device.  ``start`` enqueues, ``finish`` waits for that batch only; two
    buffer sets, so batch k+1 can be started while batch k is still in flight.  A rank whose
    count exceeds ``cap`` is reported through ``overflow`` (the caller falls back to
    ``allgather_hits`` for that batch).
    """

    def __init__(self, cap: int, device, group=None, depth: int = 2):
        self.cap, self.group = int(cap), group
        self.world = dist.get_world_size(group)
        self.send = [torch.zeros((self.cap + 1, HIT_WORDS), dtype=torch.int64, device=device) for _ in range(depth)]
        self.recv = [torch.zeros((self.world, self.cap + 1, HIT_WORDS), dtype=torch.int64, device=device)
                     for _ in range(depth)]
        self.slot = 0
        self._iota = torch.arange(self.cap, device=device, dtype=torch.int64)

    def start(self, hits: torch.Tensor, n_hits: torch.Tensor):
        """hits int64[>=?, 6] ordered records of this rank, n_hits int64[1].  Returns a handle."""
        k = self.slot
        self.slot = (self.slot + 1) % len(self.send)
        send, recv = self.send[k], self.recv[k]
        rows = min(self.cap, hits.shape[0])
        send[0, 0] = n_hits[0]
        send[1:1 + rows] = hits[:rows]
        work = dist.all_gather_into_tensor(recv.view(self.world * (self.cap + 1), HIT_WORDS), send, group=self.group,
                                           async_op=True)
        return work, k

    def finish(self, handle):
        """(merged int64[world*cap + 1, 6], total int64[1], overflow bool[1]) on the device; rows
        [0, total) of ``merged`` are the global ordered hit list, the rest is scratch."""
        work, k = handle
        work.wait()
        recv = self.recv[k]
        counts = recv[:, 0, 0]
        overflow = (counts > self.cap).any().reshape(1)
        c = counts.clamp(max=self.cap)
        starts = torch.cumsum(c, 0) - c
        total = c.sum().reshape(1)
        trash = self.world * self.cap
        idx = starts[:, None] + self._iota[None, :]
        idx = torch.where(self._iota[None, :] < c[:, None], idx, torch.full_like(idx, trash))
        merged = torch.empty((trash + 1, HIT_WORDS), dtype=torch.int64, device=recv.device)
        merged.index_copy_(0, idx.reshape(-1), recv[:, 1:, :].reshape(-1, HIT_WORDS))
        return merged, total, overflow


class RcclHitExchange:
    """The merge through the C ABI: ``gf_comm_*`` + ``gf_allgather_hits_device`` (include/gfmatch.h) — RCCL called
    by libgfmatch.so itself, which is what a Rust / C++ host binds.  ``torch.distributed`` only
    carries the 128-byte communicator id (one broadcast) — any launcher's side channel would do.

    ``start`` queues stage -> ncclAllGather -> pack on a side stream behind the work already queued on the
    caller's stream and returns; ``finish`` makes the caller's stream wait for that batch and hands out
    (merged int64[world*cap, 6], totals int64[2 + world]) — totals[0] records of ``merged`` are the global ordered
    list, totals[1] != 0 says a rank had more than ``cap`` records.  Two buffer sets: batch k+1 may be started
    while batch k is still in flight.
    """

    def __init__(self, cap: int, device, group=None, depth: int = 2):
        from . import _lib
        import ctypes as C
        self._lib, self._C = _lib, C
        L = _lib.lib()
        self.cap, self.group = int(cap), group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.device = torch.device(device)
        # Every step that can fail on one rank only is followed by an agreement over the torch process group: a rank
        # that raised alone would leave its peers inside ncclCommInitRank / the first ncclAllGather, waiting for a
        # collective it never joins.  After an agreement every rank raises, or none does.
        idt = torch.zeros(_lib.GF_COMM_ID_BYTES, dtype=torch.uint8)
        err = None
        if self.rank == 0:
            try:
                buf = (C.c_uint8 * _lib.GF_COMM_ID_BYTES)()
                _lib.check(L.gf_comm_unique_id(buf))
                idt = torch.frombuffer(bytearray(buf), dtype=torch.uint8).clone()
            except Exception as e:  # noqa: BLE001
                err = e
        self._agree(err, "gf_comm_unique_id")
        src = dist.get_global_rank(group, 0) if group is not None else 0
        if dist.get_backend(group) == "nccl":   # (the backend moves device tensors only)
            idd = idt.to(self.device)
            dist.broadcast(idd, src=src, group=group)
            idt = idd.cpu()
        else:
            dist.broadcast(idt, src=src, group=group)
        idb = (C.c_uint8 * _lib.GF_COMM_ID_BYTES).from_buffer_copy(idt.numpy().tobytes())
        h = C.c_void_p()
        try:
            _lib.check(L.gf_comm_init(idb, self.rank, self.world, self.device.index or 0, C.byref(h)))
        except Exception as e:  # noqa: BLE001
            err = e
        self._h = h if err is None else None
        self._agree(err, "gf_comm_init")
        ws = int(L.gf_allgather_workspace_bytes(self.world, self.cap))
        self.side = torch.cuda.Stream(device=self.device)
        self.sets = [dict(ws=torch.empty(ws, dtype=torch.uint8, device=self.device),
                          merged=torch.empty((max(self.world * self.cap, 1), HIT_WORDS), dtype=torch.int64, device=self.device),
                          totals=torch.zeros(2 + self.world, dtype=torch.int64, device=self.device),
                          done=torch.cuda.Event()) for _ in range(depth)]
        self.slot = 0

    def _agree(self, err, what: str) -> None:
        """all_reduce(MIN) of "this rank is fine" over the torch group; raises on EVERY rank when any rank failed."""
        on_dev = dist.get_backend(self.group) == "nccl"
        ok = torch.tensor([0 if err is not None else 1], dtype=torch.int32, device=self.device if on_dev else "cpu")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
        if int(ok.item()) == 0:
            raise RuntimeError("%s failed on %s: %s" % (what, "this rank (%d)" % self.rank if err is not None else "another rank",
                                                          err if err is not None else "see its message"))

    def comm_world(self):
        """(rank, world) as RCCL reports them from inside the communicator (gf_comm_rank)."""
        r, w = self._C.c_int32(-1), self._C.c_int32(-1)
        self._lib.check(self._lib.lib().gf_comm_rank(self._h, self._C.byref(r), self._C.byref(w)))
        return int(r.value), int(w.value)

    def first_exchange(self, hits: torch.Tensor, n_hits: torch.Tensor):
        """One exchange before anything is timed, with an agreement between queueing it and waiting for it: a rank
        that could not queue its ncclAllGather is found while the peers' kernels are still waiting for it, and every
        rank raises instead of hanging in the synchronisation.  Returns (merged, totals)."""
        err, h = None, None
        try:
            h = self.start(hits, n_hits)
        except Exception as e:  # noqa: BLE001
            err = e
        self._agree(err, "the first gf_allgather_hits_device")
        out = self.finish(h)
        torch.cuda.synchronize(self.device)
        return out

    def start(self, hits: torch.Tensor, n_hits: torch.Tensor):
        k = self.slot
        self.slot = (self.slot + 1) % len(self.sets)
        s = self.sets[k]
        cur = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(cur)
        self.side.wait_event(ready)       # the list is complete when the caller's queued work is
        self._lib.check(self._lib.lib().gf_allgather_hits_device(
            self._h, hits.data_ptr(), n_hits.data_ptr(), self.cap,   # (only the first min(count, cap) records are read)
            s["merged"].data_ptr(), s["totals"].data_ptr(), s["ws"].data_ptr(), self.side.cuda_stream))
        s["done"].record(self.side)
        s["keep"] = (hits, n_hits)        # (alive until the side stream has read them)
        return k

    def finish(self, handle):
        s = self.sets[handle]
        torch.cuda.current_stream(self.device).wait_event(s["done"])
        return s["merged"], s["totals"]

    def close(self):
        if getattr(self, "_h", None):
            torch.cuda.synchronize(self.device)
            self._lib.lib().gf_comm_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pack_gathered_hits(recv: torch.Tensor, world: int, cap: int):
    """gf_pack_gathered_hits_device on a receive buffer int64[world * (cap + 1), 6] (record 0 of every block = its
    count in word 0): (merged int64[world*cap, 6], totals int64[2 + world])."""
    from . import _lib
    merged = torch.empty((max(world * cap, 1), HIT_WORDS), dtype=torch.int64, device=recv.device)
    totals = torch.zeros(2 + world, dtype=torch.int64, device=recv.device)
    _lib.check(_lib.lib().gf_pack_gathered_hits_device(recv.data_ptr(), world, cap, merged.data_ptr(), totals.data_ptr(),
                                                      torch.cuda.current_stream(recv.device).cuda_stream))
    return merged, totals
