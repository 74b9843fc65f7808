#!/bin/bash
# rocprofv3 over tools/bench_pairs.py (GPU box): kernel trace + the two calibrated byte counters, then
# tools/stage_traffic.py -> per-stage / per-kernel bytes and GB/s.   bash tools/profile_pairs.sh <tag> [pairs]
set -o pipefail
TAG=${1:-r03}; PAIRS=${2:-10000000}
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_pairs_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $REPO/tools/bench_pairs.py --pairs $PAIRS --steps 3 --warmup 1 --profile-mode"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $B > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
for pmc in "WRITE_SIZE" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_32B_sum"; do
  name=$(echo $pmc | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc_$name -o p -- $B > $OUT/pmc_$name.log 2>&1 || { tail -5 $OUT/pmc_$name.log; exit 1; }
done
cd $REPO
python3 tools/stage_traffic.py $OUT $PAIRS $OUT/frontend_traffic.json > $OUT/stage_traffic.txt 2>&1
cat $OUT/stage_traffic.txt
find $OUT -name "*.csv" -size +512k -delete
