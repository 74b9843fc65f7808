"""Condense a tools/profile_gpu.sh output directory into a short text summary
(per-kernel stats + per-launch PMC sums for the mapping kernel)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print("# rocprofv3 summary for", os.path.basename(out))
for f in find("trace/**/*kernel_stats.csv"):
    print("\n## kernel stats (%s)" % os.path.relpath(f, out))
    rows = list(csv.DictReader(open(f)))
    for r in rows[:12]:
        print("%-70s calls=%-5s total_ns=%-14s avg_ns=%-12s pct=%s" % (
            r.get("Name", "")[:70], r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage")))
for f in find("trace/**/*kernel_trace.csv"):
    rows = list(csv.DictReader(open(f)))
    names = sorted({r.get("Kernel_Name", "").split("(")[0].replace("void ", "") for r in rows
                    if "gf_k_" in r.get("Kernel_Name", "")})
    for hot in names:
      ks = [r for r in rows if r.get("Kernel_Name", "").split("(")[0].replace("void ", "") == hot]
      if ks:
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ks]
        r0 = ks[0]
        print("\n## %s dispatches: n=%d avg=%.3f ms min=%.3f max=%.3f  VGPR=%s SGPR=%s LDS=%s grid=%s wg=%s" % (
            hot, len(d), sum(d) / len(d) / 1e6, min(d) / 1e6, max(d) / 1e6, r0.get("VGPR_Count"), r0.get("SGPR_Count"),
            r0.get("LDS_Block_Size"), r0.get("Grid_Size"), r0.get("Workgroup_Size")))
HOT = ("gf_k_",)
for f in find("pmc_*/**/*counter_collection.csv"):
    rows = list(csv.DictReader(open(f)))
    acc = defaultdict(lambda: defaultdict(list))
    for r in rows:
        kn = r.get("Kernel_Name", "")
        for h in HOT:
            if h in kn:
                short = kn.split("(")[0].replace("void ", "")
                acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kname, d in acc.items():
        print("\n## pmc (%s) %s, per launch (mean over %d launches)" % (
            os.path.relpath(f, out).split(os.sep)[0], kname, len(next(iter(d.values())))))
        for k, v in d.items():
            print("  %-28s %.6g" % (k, sum(v) / len(v)))
