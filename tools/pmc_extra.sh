#!/bin/bash
# Extra SQ counter passes (issue-side view of the kernels).  Run on the GPU box:
#   bash tools/pmc_extra.sh <tag>
set -o pipefail
TAG=${1:-x}
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmcx_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity"
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_BRANCH SQ_LEVEL_WAVES SQ_CYCLES"; do
  name=$(echo $pmc | tr ' ' '_' | cut -c1-40)
  echo "== pmc $pmc =="
  timeout -k 10 200 rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc_$name -o pmc -- $BENCH > $OUT/pmc_${name}.log 2>&1 || { echo "pmc $pmc failed"; tail -5 $OUT/pmc_${name}.log; }
done
cd $REPO
python3 tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
rm -rf $OUT/pmc_*/
