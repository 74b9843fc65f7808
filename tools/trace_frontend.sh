#!/bin/bash
# Per-kernel average durations of tools/bench_frontend.py (rocprofv3 kernel trace).
REPO=$(pwd); OUT=$REPO/gpurun_out/trace_frontend_tmp
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $REPO/tools/bench_frontend.py --steps 3 --warmup 1 --check 1000 "$@" > $REPO/gpurun_out/trace_frontend_bench.log 2>&1
cd $REPO
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/t_kernel_stats.csv")))
for r in rows:
    if "gf_k_" in r["Name"]:
        print("%-90s calls=%-4s avg_us=%9.1f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3))
PY
cp $OUT/t_kernel_stats.csv $REPO/gpurun_out/frontend_kernel_stats.csv
rm -rf $OUT
