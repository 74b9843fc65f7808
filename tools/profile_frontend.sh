#!/bin/bash
# rocprofv3 kernel trace of the front-end bench (FASTQ cutting + fast_merge); run through gpurun:
#   bash tools/profile_frontend.sh <tag>
set -o pipefail
TAG=${1:-r01_frontend}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $REPO/tools/bench_frontend.py --steps 5 --warmup 1 --check 0 > $OUT/trace_bench.log 2>&1 || { echo trace failed; tail -20 $OUT/trace_bench.log; exit 1; }
cd $REPO
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
grep "gf_k" $OUT/kernel_stats.csv | cut -c1-200
