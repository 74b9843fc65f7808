import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from genefuserust_amd import Indexer, synth
genes = synth.make_geneset("IDX-D")
ix = Indexer.from_gene_slices(genes.seqs, genes.reversed_flags); ix.make_index()
n, L = 20_000_000, 150
rb = synth.make_reads(genes, n, read_len=L, mix="PANEL", seed=1, device="cuda")
counts = torch.empty(n, dtype=torch.uint8, device="cuda"); matches = torch.empty((n, 2, 4), dtype=torch.int32, device="cuda")
for _ in range(2):
    ix.map_reads_device(rb.bases, rb.offsets, L, counts, matches); ix.compact_hits_device(counts, matches, n, cap=n // 16)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); ix.map_reads_device(rb.bases, rb.offsets, L, counts, matches); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    h = ix.compact_hits_device(counts, matches, n, cap=n // 16); t3 = time.perf_counter()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    print("map call %.3f ms, map sync %.3f ms, compact call %.3f ms, compact sync %.3f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3), flush=True)
t0 = time.perf_counter()
for _ in range(10):
    ix.map_reads_device(rb.bases, rb.offsets, L, counts, matches); ix.compact_hits_device(counts, matches, n, cap=n // 16)
torch.cuda.synchronize()
print("10 steps: %.3f ms per step" % ((time.perf_counter() - t0) * 100), flush=True)
t0 = time.perf_counter()
for _ in range(10):
    ix.map_reads_device(rb.bases, rb.offsets, L, counts, matches)
torch.cuda.synchronize()
print("10 maps only: %.3f ms per step" % ((time.perf_counter() - t0) * 100), flush=True)
# the bench's loop: two timing events per step on the current stream
stream = torch.cuda.current_stream()
for tag, make in (("events created before", True), ("no events", False)):
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)] if make else None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(10):
        if evs: evs[k][0].record(stream)
        ix.map_reads_device(rb.bases, rb.offsets, L, counts, matches)
        if evs: evs[k][1].record(stream)
        out = ix.compact_hits_device(counts, matches, n, read_id_base=0, cap=n // 16)
    torch.cuda.synchronize()
    print("%s: %.3f ms per step" % (tag, (time.perf_counter() - t0) * 100), flush=True)
    if evs:
        print("   event-timed map: %.3f ms" % (sum(a.elapsed_time(b) for a, b in evs) / 10))
import torch.distributed as dist  # noqa
t0 = time.perf_counter()
for k in range(10):
    ix.map_reads_device(rb.bases, rb.offsets, L, counts, matches)
    out = ix.compact_hits_device(counts, matches, n, read_id_base=0, cap=n // 16)
torch.cuda.synchronize()
print("after importing torch.distributed: %.3f ms per step" % ((time.perf_counter() - t0) * 100), flush=True)
