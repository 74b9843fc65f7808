import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from genefuserust_amd import Indexer
for genes in ([], [b""], [b"ACGT"], [b"N" * 100], [b"ACGTACGTACGTACGTA"], [b"ACGTTGCAAGCTTAGC" * 3, b"", b"acgt" * 10]):
    ix = Indexer.from_gene_slices(genes)
    ix.make_index()
    info = ix.info()
    reads = [b"ACGTTGCAAGCTTAGC" * 5, b"", b"A" * 150]
    out = ix.map_reads(reads)
    print(len(genes), [len(g) for g in genes], "keys", info["n_keys"], "sites", info["n_sites"], "results", [len(x) for x in out])
    ix.close()
print("OK")
