#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point gf_map_reads_hits (never the bench value):
reads in host memory -> H2D -> map -> compact -> D2H of the hit records."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from genefuserust_amd import Indexer  # noqa: E402
from genefuserust_amd import synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
    genes = synth.make_geneset("IDX-D")
    ix = Indexer.from_gene_slices(genes.seqs, genes.reversed_flags)
    ix.make_index()
    rb = synth.make_reads(genes, n, read_len=150, mix="PANEL", seed=1, device="cuda")
    bases = rb.bases.cpu().numpy()
    offsets = rb.offsets.cpu().numpy()
    out = {}
    for kind in ("pageable", "pinned"):
        if kind == "pinned":
            pb = torch.empty(bases.size, dtype=torch.uint8).pin_memory()
            pb.numpy()[:] = bases
            po = torch.empty(offsets.size, dtype=torch.int64).pin_memory()
            po.numpy()[:] = offsets
            b, o = pb.numpy(), po.numpy()
        else:
            b, o = bases, offsets
        ix.map_reads_hits(b, o)  # warm-up (workspace, first-touch)
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            hits = ix.map_reads_hits(b, o)
        dt = (time.perf_counter() - t0) / reps
        out[kind] = {"ms": dt * 1e3, "reads_per_s": n / dt, "host_GBps": (bases.nbytes + offsets.nbytes) / dt / 1e9,
                     "hits": int(len(hits))}
    print(json.dumps({"n_reads": n, "entry": "gf_map_reads_hits", **out}))


if __name__ == "__main__":
    main()
