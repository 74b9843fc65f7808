#!/usr/bin/env python3
"""Per-stage and per-kernel fabric bytes of tools/bench_pairs.py from a rocprofv3 run (tools/profile_pairs.sh):

    python tools/stage_traffic.py gpurun_out/prof_pairs_<tag> <pairs> [out.json]

The dispatches of the library's kernels are cut into stage invocations by their first kernel — gf_k_fq_count opens a
FASTQ cut, gf_k_merge_find_* a gf_scan_pairs_device call — in dispatch order (the same in every rocprofv3 pass: one
process each, deterministic launch sequence).  Bytes as calibrated in r03: read = 32 B x TCC_EA0_RDREQ_DRAM_32B_sum,
written = 1024 x WRITE_SIZE.  Durations from the kernel trace."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(kn):
    return kn.split("(")[0].replace("void ", "").strip()


def ordered(path_glob, value_of):
    rows = []
    for f in glob.glob(path_glob, recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows


def main():
    out, pairs = sys.argv[1], int(sys.argv[2])
    tr = [r for r in ordered(out + "/trace/**/*kernel_trace.csv", None) if "gf_k_" in r["Kernel_Name"]]
    tr.sort(key=lambda r: int(r["Start_Timestamp"]))
    seq = [(short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6) for r in tr]
    counters = {}
    for d in glob.glob(out + "/pmc_*"):
        if not os.path.isdir(d):
            continue
        per = defaultdict(dict)
        for r in ordered(d + "/**/*counter_collection.csv", None):
            if "gf_k_" in r["Kernel_Name"]:
                per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
                per[int(r["Dispatch_Id"])]["_name"] = short(r["Kernel_Name"])
        lst = [per[k] for k in sorted(per)]
        for cn in {c for p in lst for c in p if c != "_name"}:
            counters[cn] = lst
    n = len(seq)
    for cn, lst in counters.items():
        if len(lst) != n or any(a["_name"] != b[0] for a, b in zip(lst, seq)):
            raise SystemExit("dispatch sequences of the trace and of the %s pass differ (%d vs %d)" % (cn, n, len(lst)))
    rd = [32 * counters["TCC_EA0_RDREQ_DRAM_32B_sum"][i].get("TCC_EA0_RDREQ_DRAM_32B_sum", 0) for i in range(n)]
    wr = [1024 * counters["WRITE_SIZE"][i].get("WRITE_SIZE", 0) for i in range(n)]
    # stage invocations
    stages, cur = [], None
    for i, (name, ms) in enumerate(seq):
        if name.startswith("gf_k_fq_count"):
            cur = {"stage": "fastq_cut", "k": []}
            stages.append(cur)
        elif name.startswith("gf_k_merge_find"):
            cur = {"stage": "scan_pairs_device", "k": []}
            stages.append(cur)
        elif name.startswith(("gf_k_index_", "gf_k_classify", "gf_k_sort_dupes", "gf_k_filter_", "gf_k_upper_", "gf_k_pair_hits_finish")):
            cur = None   # (the index build; the device tail, which bench_pairs.py times as a stage of its own)
        if cur is not None:
            cur["k"].append(i)
    # (bench_pairs.py leaves the qualities in the text unless --full-gather: tools/profile_pairs.sh runs it plainly)
    res = {"pairs": pairs, "stages": {}, "source_dir": os.path.basename(out), "lean": True}
    for st in ("fastq_cut", "scan_pairs_device"):
        inv = [s for s in stages if s["stage"] == st]
        if not inv:
            continue
        k = len(inv)
        ms = sum(seq[i][1] for s in inv for i in s["k"]) / k
        r = sum(rd[i] for s in inv for i in s["k"]) / k
        w = sum(wr[i] for s in inv for i in s["k"]) / k
        byk = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
        for s in inv:
            for i in s["k"]:
                e = byk[seq[i][0]]
                e[0] += 1; e[1] += seq[i][1]; e[2] += rd[i]; e[3] += wr[i]
        res["stages"][st] = {
            "invocations_profiled": k, "kernel_ms": ms, "read_bytes": r, "write_bytes": w,
            "GBps": (r + w) / ms / 1e6, "frac_of_8TBps": (r + w) / ms / 1e6 / 8000.0,
            "kernels": {kn: {"launches": e[0] / k, "ms": e[1] / k, "read_bytes": e[2] / k, "write_bytes": e[3] / k,
                             "GBps": (e[2] + e[3]) / e[1] / 1e6 if e[1] else None} for kn, e in sorted(byk.items(), key=lambda kv: -kv[1][1])}}
    path = sys.argv[3] if len(sys.argv) > 3 else None
    if path:
        json.dump(res, open(path, "w"), indent=1)
    for st, e in res["stages"].items():
        print("## %s: %.3f ms of kernels, read %.3f GB, written %.3f GB: %.0f GB/s = %.2f of 8 TB/s (%d invocations)" % (
            st, e["kernel_ms"], e["read_bytes"] / 1e9, e["write_bytes"] / 1e9, e["GBps"], e["frac_of_8TBps"], e["invocations_profiled"]))
        for kn, v in e["kernels"].items():
            print("   %-46s x%-4.1f %7.3f ms  read %7.3f GB  written %7.3f GB  %6.0f GB/s" % (
                kn[:46], v["launches"], v["ms"], v["read_bytes"] / 1e9, v["write_bytes"] / 1e9, v["GBps"] or 0))


if __name__ == "__main__":
    main()
