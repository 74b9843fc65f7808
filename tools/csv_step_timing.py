"""Where a multi-CSV step's time goes (BASELINE configs[4]): lap times per CSV of the pieces
bench.py --config 4 runs — index object, make_index, mapping, compaction, the count's read-back, close.
    python3 tools/csv_step_timing.py [n_reads] [n_csv]
"""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
from genefuserust_amd import Indexer, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
n_csv = int(sys.argv[2]) if len(sys.argv) > 2 else 8
L = 150
shapes = ["IDX-C" if k % 2 == 0 else "IDX-D" for k in range(n_csv)]
sets = [synth.make_geneset(shapes[k], seed=1000 + 37 * k) for k in range(n_csv)]
half = n // 2
ra = synth.make_reads(sets[0], half, read_len=L, mix="PANEL", seed=1, device="cuda")
rb = synth.make_reads(sets[1], n - half, read_len=L, mix="PANEL", seed=2, device="cuda")
bases = torch.cat([ra.bases, rb.bases])
offsets = torch.arange(n + 1, device="cuda", dtype=torch.int64) * L
del ra, rb


def step(show):
    packed = None
    tot = {}
    t_step = time.perf_counter()
    for k in range(n_csv):
        laps = []
        t = time.perf_counter()

        def lap(name, sync=True):
            nonlocal t
            if sync:
                torch.cuda.synchronize()
            t2 = time.perf_counter()
            laps.append((name, 1e3 * (t2 - t)))
            t = t2
        gs = sets[k]
        ix = Indexer.from_gene_slices(gs.seqs, gs.reversed_flags)
        lap("object")
        ix.make_index()
        lap("make_index")
        if packed is None:
            packed = ix.pack_bases_device(bases)
            lap("pack")
        counts, matches = ix.map_reads_packed_device(packed[0], packed[1], offsets, L)
        lap("map")
        hits, n_hits = ix.compact_hits_device(counts, matches, n, cap=max(n // 8, 4096))
        lap("compact")
        kk = int(n_hits.item())
        lap("count")
        ix.close()
        lap("close")
        del counts, matches, hits, n_hits
        lap("del")
        for name, ms in laps:
            tot[name] = tot.get(name, 0.0) + ms
        if show:
            print("csv %2d %s: " % (k, shapes[k]) + "  ".join("%s %.2f" % x for x in laps) + "  hits %d" % kk, flush=True)
    torch.cuda.synchronize()
    if show:
        print("step %.1f ms; sums: " % (1e3 * (time.perf_counter() - t_step)) + "  ".join("%s %.1f" % x for x in tot.items()), flush=True)


step(False)
step(True)
step(True)
