#!/usr/bin/env python3
"""Pairs per second from FASTQ text in (pinned) HOST memory to the hit list: raw text chunks over the link,
records + scan_pair_end policy on the device, chunk k+1's copy overlapping chunk k's kernels
(genefuserust_amd/scan_stream.py), the host-side tail included.  One JSON line; the link's own rate for
one large pinned copy beside it."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tools.bench_frontend import make_text  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=4_000_000)
    ap.add_argument("--chunk-mb", type=int, default=256)
    a = ap.parse_args()
    from genefuserust_amd import Indexer, synth
    from genefuserust_amd.scan_stream import scan_pair_end_text, scan_pair_text_stream
    from genefuserust_amd.stream import pinned_empty
    dev = torch.device("cuda", 0)
    genes = synth.make_geneset("IDX-D")
    ix = Indexer.from_gene_slices(genes.seqs, genes.reversed_flags)
    ix.make_index()
    n, L = a.pairs, 150
    pr = synth.make_pairs(genes, n, read_len=L, seed=20240302, device="cuda")
    texts = []
    for mate, (b, q) in enumerate(((pr.l_bases, pr.l_quals), (pr.r_bases, pr.r_quals)), 1):
        t = make_text(b, q, n, L, mate, dev)
        h = pinned_empty(t.numel(), np.uint8)
        h[:] = t.cpu().numpy()
        texts.append(h)
        del t
    del pr
    torch.cuda.empty_cache()
    out = {}
    for rep in range(2):   # the first pass warms the arenas
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        found, counters = scan_pair_end_text(ix, texts[0], texts[1], chunk_bytes=a.chunk_mb << 20)
        dt = time.perf_counter() - t0
    # the stream alone (records and their reads on the host), without the Python objects of the tail
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_rec = sum(t[0].shape[0] for t in scan_pair_text_stream(ix, texts[0], texts[1], chunk_bytes=a.chunk_mb << 20))
    dt_stream = time.perf_counter() - t0
    nbytes = int(texts[0].size + texts[1].size)
    d = torch.empty(texts[0].size, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    d.copy_(torch.from_numpy(texts[0]), non_blocking=True)
    torch.cuda.synchronize()
    link = texts[0].size / (time.perf_counter() - t0) / 1e9
    print(json.dumps({
        "metric": "read pairs per second from FASTQ text in pinned host memory to the ReadMatch list (streamed chunks)",
        "value": n / dt_stream, "unit": "pairs/s", "pairs": n, "seconds": dt_stream, "text_bytes": nbytes,
        "host_text_GBps": nbytes / dt_stream / 1e9, "records": n_rec,
        "with_python_readmatch_objects": {"pairs_per_s": n / dt, "seconds": dt},
        "link_GBps_one_copy": link, "chunk_mb": a.chunk_mb, "counters": counters, "matches": len(found)}))


if __name__ == "__main__":
    main()
