#!/bin/bash
# Per-kernel average durations of one bench configuration (rocprofv3 kernel trace).
#   bash tools/trace_kernels.sh [bench args]
REPO=$(pwd); OUT=$REPO/gpurun_out/trace_tmp
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-parity "$@" > /dev/null 2>&1
cd $REPO
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/t_kernel_stats.csv")))
for r in rows:
    if "gf_k_" in r["Name"]:
        print("%-62s calls=%-3s avg_us=%9.1f" % (r["Name"][:62], r["Calls"], float(r["AverageNs"])/1e3))
PY
rm -rf $OUT
