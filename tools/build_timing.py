import os, time, sys
sys.path.insert(0, "/root/repo")
os.environ["GF_BUILD_TIMING"] = "1"
import torch
from genefuserust_amd import Indexer, synth
for shape in ("IDX-C", "IDX-D", "IDX-C"):
    g = synth.make_geneset(shape)
    t0 = time.perf_counter(); ix = Indexer.from_gene_slices(g.seqs, g.reversed_flags); t1 = time.perf_counter()
    ix.make_index(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ix.close(); torch.cuda.synchronize(); t3 = time.perf_counter()
    print(shape, "from_gene_slices %.2f ms, make_index %.2f ms, close %.2f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3), flush=True)
