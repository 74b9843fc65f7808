"""BASELINE configs[4] shape: the index is rebuilt per CSV (multi-CSV mode,
fusion_scan.rs:62-188; benchmark_res/hg38_fusion_csv_list.txt alternates the cancer
and druggable lists 8 times).  Times 16 make_index() calls alternating IDX-C / IDX-D."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genefuserust_amd import Indexer, synth

sets = {k: synth.make_geneset(k) for k in ("IDX-C", "IDX-D")}
times = []
for it in range(16):
    name = "IDX-C" if it % 2 == 0 else "IDX-D"
    g = sets[name]
    ix = Indexer.from_gene_slices(g.seqs, g.reversed_flags)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ix.make_index()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    times.append((name, dt, ix.info()["n_keys"]))
    ix.close()
for name, dt, nk in times:
    print("%s make_index %.1f ms (%d keys)" % (name, dt * 1e3, nk))
print("total %.3f s for 16 rebuilds" % sum(t for _, t, _ in times))
