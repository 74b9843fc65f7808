#!/bin/bash
# kernel trace of the pair pipeline bench (gpurun, from the repo root): bash tools/trace_pairs.sh <tag>
set -o pipefail
TAG=${1:-r02}
REPO=$(pwd)
OUT=$REPO/gpurun_out/pairs_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $REPO/tools/bench_pairs.py --steps 3 --check 500 > $OUT/bench.log 2>&1 || { echo trace failed; tail -20 $OUT/bench.log; exit 1; }
cd $REPO
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/kernel_stats.csv")))
for r in rows:
    if "gf_k_" in r["Name"]:
        print("%-60s calls=%-4s avg_ms=%8.3f total_ms=%8.3f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e6, float(r["TotalDurationNs"])/1e6))
PY
tail -1 $OUT/bench.log
