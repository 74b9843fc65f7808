#!/bin/bash
# kernel trace + L2 hit/miss counters per dispatch of one bench configuration (GPU box):
#   bash tools/trace_cfg.sh <tag> <bench args...>     (environment switches pass through)
set -o pipefail
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/trace_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $REPO/bench.py --steps 3 --warmup 1 --profile-mode $*"
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- $B > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc -o p -- $B > $OUT/pmc.log 2>&1 || { tail -5 $OUT/pmc.log; }
python3 - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows = [r for r in rows if "gf_k_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
pm = {}
for f in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gf_k_" in r["Kernel_Name"]:
            pm.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
pml = [pm[k] for k in sorted(pm)]
# the last pass: from the last seed+verify on
last = max(i for i, r in enumerate(rows) if "seedverify" in r["Kernel_Name"])
names = [r["Kernel_Name"].split("(")[0].replace("void ", "") for r in rows]
# align the pmc dispatch list with the trace by order of the gf_k_ kernels
for i in range(last, len(rows)):
    r = rows[i]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    c = pml[i] if i < len(pml) else {}
    print("%-44s %8.3f ms   L2 hits %7.1f M  misses %7.1f M" % (names[i][:44], d, c.get("TCC_HIT_sum", 0) / 1e6, c.get("TCC_MISS_sum", 0) / 1e6))
PY
