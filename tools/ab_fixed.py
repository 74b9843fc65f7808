import sys, torch
sys.path.insert(0, ".")
from genefuserust_amd import Indexer, synth
genes = synth.make_geneset("IDX-D")
ix = Indexer.from_gene_slices(genes.seqs, genes.reversed_flags); ix.make_index()
n, L = 20_000_000, 150
reads = synth.make_pair_reads(genes, n // 2, read_len=L, seed=20240116, device="cuda")
c = torch.empty(n, dtype=torch.uint8, device="cuda"); m = torch.empty((n, 2, 4), dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream()
def t(fn, k=8):
    fn(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(k): fn()
    e1.record(st); torch.cuda.synchronize(); return e0.elapsed_time(e1) / k
for rep in range(4):
    a = t(lambda: ix.map_reads_device(reads.bases, reads.offsets, L, c, m))
    b = t(lambda: ix.map_reads_fixed_device(reads.bases, L, c, m))
    print("offsets %.4f ms  fixed %.4f ms  (%.2f %%)" % (a, b, 100 * (b - a) / a))
