#!/bin/bash
# Per-kernel average durations of tools/bench_merge.py (rocprofv3 kernel trace).
REPO=$(pwd); OUT=$REPO/gpurun_out/trace_merge_tmp
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $REPO/tools/bench_merge.py --steps 5 --warmup 1 --check 2000 "$@" > $REPO/gpurun_out/trace_merge_bench.log 2>&1
cd $REPO
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/t_kernel_stats.csv")))
for r in rows[:14]:
    print("%-80s calls=%-4s avg_us=%9.1f pct=%s" % (r["Name"][:80], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
cp $OUT/t_kernel_stats.csv $REPO/gpurun_out/merge_kernel_stats.csv
rm -rf $OUT
