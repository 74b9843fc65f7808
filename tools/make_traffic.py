#!/usr/bin/env python3
"""profiles/hbm_traffic.json from a tools/profile_gpu.sh output directory (r03: calibrated; r04: every entry stamped with
the commit, the hash of the kernel sources and the hash of the library the counters were taken from).

    python tools/make_traffic.py gpurun_out/prof_<tag> <key> <source-name>
    python tools/make_traffic.py gpurun_out/prof_<tag> <key> <source-name> --per-step N
        (every gf_k_* kernel of the run, divided by its N steps: multi-CSV mode, where a step is 16 index
         rebuilds + 16 mapping passes + one packing of the reads)

What r03's calibration kernels (tools/gf_calib.hip, launched by bench.py --calib in the same rocprofv3 passes)
established on gfx950 / ROCm 7.2:
  * every read request the L2 sends to the fabric is a 128-byte line — streams of 16 and of 4 bytes per lane and
    scattered dword loads alike: TCC_EA0_RDREQ_DRAM_32B_sum / TCC_EA0_RDREQ_DRAM_sum = 4.00;
  * TCC_EA0_RDREQ_DRAM_32B_sum x 32 B reproduces the known byte count of both streams exactly;
  * FETCH_SIZE tallies each of those requests at 64 B (TCC_BUBBLE_sum, its count of 128-byte requests, reads 0 on
    gfx950): exactly half of the bytes, for every kernel here, not only for wide streams.
So: read bytes = 32 x TCC_EA0_RDREQ_DRAM_32B_sum, write bytes = 1024 x WRITE_SIZE (KiB), per kernel, summed over a
pass's kernels.  Infinity-Cache hits are included (these are L2 <-> fabric counters).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

PASS_KERNELS = ("gf_k_seedverify_stream", "gf_k_probe_filter", "gf_k_probe_buckets", "gf_k_map_reads_list")
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def build_stamp(out_dir: str) -> dict:
    """Which build the counters describe.  tools/profile_gpu.sh writes build_stamp.json next to the
    counter files ON THE GPU BOX (hash of the sources it ran: bench.kernel_source_sha, hash of the libgfmatch.so it
    loaded); the commit is added here, in the container, where .git is — and refused when the tree's sources are not
    the ones that were profiled."""
    import subprocess
    import sys
    sys.path.insert(0, ROOT)
    import bench
    here = {"kernel_src_sha": bench.kernel_source_sha(), "lib_sha": bench.library_sha()}
    path = os.path.join(out_dir, "build_stamp.json")
    if not os.path.exists(path):
        raise SystemExit("%s is missing: profile with tools/profile_gpu.sh (it writes the stamp on the GPU box)" % path)
    st = json.load(open(path))
    if st["kernel_src_sha"] != here["kernel_src_sha"]:
        raise SystemExit("the profile was taken from sources %s, the tree holds %s: profile again" % (st["kernel_src_sha"], here["kernel_src_sha"]))
    try:
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True, check=True).stdout.strip()
        dirty = bool(subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "genefuserust_amd/csrc", "include/gfmatch.h"],
                                    capture_output=True, text=True).stdout.strip())
    except Exception:
        # the GPU box has no .git: tools/stamp_head.sh wrote the commit into .git_head before the snapshot was taken
        head, dirty = None, None
        try:
            gh = json.load(open(os.path.join(ROOT, ".git_head")))
            if gh.get("kernel_src_sha") == st["kernel_src_sha"]:
                head, dirty = gh.get("git_head"), gh.get("git_dirty_csrc")
        except Exception:
            pass
    return {"git_head": head, "git_dirty_csrc": dirty, "kernel_src_sha": st["kernel_src_sha"], "lib_sha": st["lib_sha"],
            "bench_args": st.get("bench_args")}


def short(kn: str) -> str:
    return kn.split("(")[0].replace("void ", "").strip()


ROUND = os.environ.get("GF_ROUND", "r04")


def main():
    out, key, source = sys.argv[1], sys.argv[2], sys.argv[3]
    sums = defaultdict(lambda: defaultdict(float))   # kernel -> counter -> sum over dispatches
    nd = defaultdict(lambda: defaultdict(int))
    for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            sums[k][r["Counter_Name"]] += float(r["Counter_Value"])
            nd[k][r["Counter_Name"]] += 1
    if "--per-step" in sys.argv:
        steps = int(sys.argv[sys.argv.index("--per-step") + 1])
        per_kernel, rd_tot, wr_tot = {}, 0.0, 0.0
        for k in sorted(sums):
            if not k.startswith("gf_k_") or k.startswith("gf_k_calib"):
                continue
            rd = 32 * sums[k].get("TCC_EA0_RDREQ_DRAM_32B_sum", 0.0) / steps
            wr = 1024 * sums[k].get("WRITE_SIZE", 0.0) / steps
            per_kernel[k] = {"launches_per_step": nd[k].get("WRITE_SIZE", 0) / steps, "read_bytes": int(rd), "write_bytes": int(wr)}
            rd_tot += rd
            wr_tot += wr
        entry = {"round": ROUND, **build_stamp(out), "hbm_bytes_per_launch": int(rd_tot + wr_tot), "read_bytes_per_launch": int(rd_tot),
                 "write_bytes_per_launch": int(wr_tot), "steps_profiled": steps, "per_kernel": per_kernel,
                 "unit_of_launch": "one step of the workload (all its kernels)",
                 "correction": "read bytes = 32 B x TCC_EA0_RDREQ_DRAM_32B_sum; write bytes = WRITE_SIZE", "source": source}
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "hbm_traffic.json")
        tj = json.load(open(path))
        tj[key] = entry
        json.dump(tj, open(path, "w"), indent=1)
        print(json.dumps({k: v for k, v in entry.items() if k != "per_kernel"}, indent=1))
        for k, e in sorted(per_kernel.items(), key=lambda kv: -kv[1]["read_bytes"] - kv[1]["write_bytes"])[:12]:
            print("%-50s x%.1f read %.3f GB write %.3f GB" % (k, e["launches_per_step"], e["read_bytes"] / 1e9, e["write_bytes"] / 1e9))
        return
    sv = [k for k in sums if k.startswith("gf_k_seedverify_stream")]
    if not sv:
        raise SystemExit("no seed+verify dispatches in %s" % out)
    # one variant of seed+verify per profile (bench.py --profile-mode): its dispatch count = the passes profiled
    main_sv = max(sv, key=lambda k: nd[k].get("WRITE_SIZE", 0))
    per_kernel, rd_tot, wr_tot = {}, 0.0, 0.0
    for k in sorted(sums):
        if not k.startswith(PASS_KERNELS) or (k.startswith("gf_k_seedverify_stream") and k != main_sv):
            continue
        c = sums[k]
        passes = {cn: nd[main_sv][cn] for cn in nd[main_sv]}
        def per_pass(cn):
            return c.get(cn, 0.0) / passes[cn] if passes.get(cn) else None
        rd32, wr = per_pass("TCC_EA0_RDREQ_DRAM_32B_sum"), per_pass("WRITE_SIZE")
        e = {
            "launches_per_pass": nd[k]["WRITE_SIZE"] / passes["WRITE_SIZE"] if passes.get("WRITE_SIZE") else None,
            "read_bytes": None if rd32 is None else int(32 * rd32),
            "write_bytes": None if wr is None else int(1024 * wr),
            "fetch_size_bytes_as_reported": None if per_pass("FETCH_SIZE") is None else int(1024 * per_pass("FETCH_SIZE")),
            "ea_read_requests": per_pass("TCC_EA0_RDREQ_sum"),
            "ea_read_requests_dram_32B_units": rd32,
            "l2_requests": per_pass("TCC_REQ_sum"), "l2_hits": per_pass("TCC_HIT_sum"), "l2_misses": per_pass("TCC_MISS_sum"),
        }
        per_kernel[k] = e
        rd_tot += e["read_bytes"] or 0
        wr_tot += e["write_bytes"] or 0
    calib = {}
    for k in sorted(sums):
        if k.startswith("gf_k_calib_stream"):
            c = sums[k]
            calib[k] = {"fetch_size_bytes_as_reported": int(1024 * c.get("FETCH_SIZE", 0)),
                        "dram_32B_units_x32_bytes": int(32 * c.get("TCC_EA0_RDREQ_DRAM_32B_sum", 0)),
                        "ea_read_requests": c.get("TCC_EA0_RDREQ_sum"), "tcc_bubble": c.get("TCC_BUBBLE_sum")}
    entry = {
        "round": ROUND,
        **build_stamp(out),
        "hbm_bytes_per_launch": int(rd_tot + wr_tot),
        "read_bytes_per_launch": int(rd_tot), "write_bytes_per_launch": int(wr_tot),
        "passes_profiled": nd[main_sv].get("WRITE_SIZE"),
        "per_kernel": per_kernel,
        "calibration": calib,
        "correction": "read bytes = 32 B x TCC_EA0_RDREQ_DRAM_32B_sum (every EA read request is a 128-byte line: 4 units); "
                      "FETCH_SIZE as reported is half of that on gfx950 (TCC_BUBBLE reads 0); write bytes = WRITE_SIZE",
        "source": source,
    }
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "hbm_traffic.json")
    tj = json.load(open(path))
    tj[key] = entry
    json.dump(tj, open(path, "w"), indent=1)
    print(json.dumps({k: v for k, v in entry.items() if k not in ("per_kernel",)}, indent=1))
    for k, e in per_kernel.items():
        print("%-46s x%.1f read %.3f GB write %.3f GB  L2 req %.1f M hits %.1f M misses %.1f M" % (
            k, e["launches_per_pass"] or 0, (e["read_bytes"] or 0) / 1e9, (e["write_bytes"] or 0) / 1e9,
            (e["l2_requests"] or 0) / 1e6, (e["l2_hits"] or 0) / 1e6, (e["l2_misses"] or 0) / 1e6))


if __name__ == "__main__":
    main()
