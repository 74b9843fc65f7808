#!/usr/bin/env python3
"""The boundary at the reference's granularity: packs of P pairs x T host threads through the C ABI.

    python tools/bench_pack_sweep.py [--pairs 4000000] [--seconds 0.4] [--out gpurun_out/pack_sweep.json] [-- extra pack_sweep args]

Writes the IDX-D gene set and a PANEL pair batch (the headline's generator) to /tmp, compiles
tools/pack_sweep.cpp against libgfmatch.so (g++: the program sees nothing but include/gfmatch.h) and runs it.  The host
loop is C++ because 16 Python threads through ctypes would measure the interpreter lock, not the library.
The reference: PACK_SIZE = 1000 pairs (common.rs:23), one Indexer::map_read per read from t-1 consumer threads
(pescanner.rs:296-311, 374-518).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build_exe(out_dir: str) -> str:
    exe = os.path.join(out_dir, "pack_sweep")
    libdir = os.path.join(ROOT, "genefuserust_amd")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "pack_sweep.cpp"),
                    "-o", exe, "-L" + libdir, "-lgfmatch", "-Wl,-rpath," + libdir, "-lpthread"], check=True)
    return exe


def write_files(out_dir: str, gene_seqs, bases, offsets):
    """genes.bin / reads.bin as tools/pack_sweep.cpp reads them (numpy arrays: uint8 bases, int64 offsets)."""
    import numpy as np
    gpath, rpath = os.path.join(out_dir, "genes.bin"), os.path.join(out_dir, "reads.bin")
    with open(gpath, "wb") as f:
        f.write(np.int32(len(gene_seqs)).tobytes())
        f.write(np.array([len(s) for s in gene_seqs], dtype=np.int64).tobytes())
        for s in gene_seqs:
            f.write(s)
    with open(rpath, "wb") as f:
        f.write(np.int64(offsets.size - 1).tobytes())
        f.write(np.ascontiguousarray(offsets, dtype=np.int64).tobytes())
        f.write(np.ascontiguousarray(bases, dtype=np.uint8).tobytes())
    return gpath, rpath


def run_sweep(gene_seqs, bases, offsets, seconds: float = 0.4, extra=(), timeout: float = 600.0) -> dict:
    """Compile tools/pack_sweep.cpp, run it on the given genes and reads in a child process (its own HIP runtime;
    the caller may have touched the GPU), return its JSON."""
    with tempfile.TemporaryDirectory(prefix="packsweep_", dir="/tmp") as d:
        exe = build_exe(d)
        gpath, rpath = write_files(d, gene_seqs, bases, offsets)
        p = subprocess.run([exe, gpath, rpath, str(seconds)] + list(extra), stdout=subprocess.PIPE, text=True, timeout=timeout)
        if p.returncode != 0:
            raise RuntimeError("pack_sweep failed (exit %d)" % p.returncode)
    return json.loads(p.stdout.strip().splitlines()[-1])


def table(j: dict) -> str:
    """rows = pack size, columns = threads, per entry and memory kind"""
    out = []
    for entry in sorted({c["entry"] for c in j["cells"]}):
        for mem in sorted({c["mem"] for c in j["cells"]}):
            cells = [c for c in j["cells"] if c["entry"] == entry and c["mem"] == mem]
            ths = sorted({c["threads"] for c in cells})
            out.append("\n%s, %s memory: M reads/s (us per call)" % (entry, mem))
            out.append("%10s " % "pairs" + " ".join("%18s" % ("%d thr" % t) for t in ths))
            for pp in sorted({c["pack_pairs"] for c in cells}):
                row = {c["threads"]: c for c in cells if c["pack_pairs"] == pp}
                out.append("%10d " % pp + " ".join("%9.1f (%6.0f)" % (row[t]["reads_per_s"] / 1e6, row[t]["us_per_call"]) if t in row else " " * 18
                                                   for t in ths))
    out.append("\nsingle-read call: %.1f us" % j["us_per_single_read_call"])
    return "\n".join(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=4_000_000)
    ap.add_argument("--seconds", type=float, default=0.4)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "pack_sweep.json"))
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("rest", nargs="*", help="extra arguments for pack_sweep (after --)")
    args = ap.parse_args()
    from genefuserust_amd import synth
    genes = synth.make_geneset("IDX-D")
    reads = synth.make_pair_reads(genes, args.pairs, read_len=150, mix="PANEL", seed=20240116, device=args.device)
    j = run_sweep(genes.seqs, reads.bases.cpu().numpy(), reads.offsets.cpu().numpy(), args.seconds, args.rest)
    j["workload"] = "%d PANEL pairs (%d reads of 150 bases) vs IDX-D, SURVEY.md 8(d)" % (args.pairs, 2 * args.pairs)
    j["env"] = {k: v for k, v in os.environ.items() if k.startswith("GF_")}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(j, open(args.out, "w"), indent=1)
    print(table(j))


if __name__ == "__main__":
    main()
