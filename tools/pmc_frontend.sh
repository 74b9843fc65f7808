#!/bin/bash
# SQ counters of the front-end kernels (FASTQ cutting, fast_merge); run through gpurun:
#   bash tools/pmc_frontend.sh <tag>
set -o pipefail
TAG=${1:-r01_frontend}
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $pmc | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $pmc --output-format csv -d $OUT/$name -o pmc -- python3 $REPO/tools/bench_frontend.py --steps 2 --warmup 1 --check 0 > $OUT/$name.log 2>&1 || { echo "pmc $pmc failed"; tail -5 $OUT/$name.log; }
done
cd $REPO
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/*counter_collection.csv") + glob.glob(out + "/*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:40]
        if "gf_k_merge" in k or "gf_k_fq" in k:
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(agg):
    print(k, {c: round(sum(v) / len(v)) for c, v in sorted(agg[k].items())})
PY
