#!/bin/bash
# Per-kernel durations of 16 alternating IDX-C / IDX-D rebuilds (rocprofv3 kernel trace of tools/bench_index_build.py).
#   bash tools/trace_build.sh [tag]      -> gpurun_out/build_<tag>_kernels.txt
REPO=$(pwd); TAG=${1:-x}; OUT=$REPO/gpurun_out/trace_build_tmp
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $REPO/tools/bench_index_build.py > $REPO/gpurun_out/build_${TAG}_bench.txt 2>&1
cd $REPO
python3 - <<PY | tee gpurun_out/build_${TAG}_kernels.txt
import csv, glob
from collections import defaultdict
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
big, small = defaultdict(list), defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("void ", "").split("(")[0]
    if not n.startswith("gf_k_"): continue
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    big[n].append(d)
for n, d in sorted(big.items(), key=lambda kv: -sum(kv[1])):
    d = sorted(d); h = len(d) // 2
    print("%-28s n=%-3d  smaller half avg %8.1f us   larger half avg %8.1f us" % (n, len(d), sum(d[:h]) / max(h, 1) / 1e3, sum(d[h:]) / max(len(d) - h, 1) / 1e3))
PY
tail -3 gpurun_out/build_${TAG}_bench.txt
rm -rf $OUT
