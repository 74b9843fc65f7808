#!/bin/bash
# PMC passes over the pair pipeline bench (gpurun, from the repo root): bash tools/pmc_pairs.sh <tag>
set -o pipefail
TAG=${1:-r02}
REPO=$(pwd)
KEEP=$REPO/gpurun_out/pmc_pairs_$TAG
OUT=/tmp/pmc_pairs_$TAG
mkdir -p $OUT $KEEP
cd /tmp && export TMPDIR=/tmp
i=0
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" \
           "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc_$i -o pmc -- python3 $REPO/tools/bench_pairs.py --steps 2 --check 200 > $OUT/pmc_$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/pmc_$i.log; }
done
cd $REPO
python3 tools/summarize_profile.py $OUT > $KEEP/summary.txt 2>&1
grep -n "gf_k_merge_write\|gf_k_merge_find" -A9 $KEEP/summary.txt
