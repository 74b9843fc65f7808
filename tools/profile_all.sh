#!/bin/bash
# The round's counter measurements at the bench's own sizes (GPU box): configs 1, 2, 3 (200 M reads each) and 4, each through
# tools/profile_gpu.sh + tools/make_traffic.py; results under gpurun_out/prof_<tag>_cfg*/ and the updated
# profiles/hbm_traffic.json copied to gpurun_out/.  Run tools/stamp_head.sh in the container first (the box has no .git).
#   bash tools/profile_all.sh <tag> [configs, default "1 2 3 4"]
set -o pipefail
TAG=${1:-r04}
CFGS=${2:-"1 2 3 4"}
export GF_ROUND=$TAG
for c in $CFGS; do
  case $c in
    1) PMC_SETS=traffic BENCH_ARGS="--calib" bash tools/profile_gpu.sh ${TAG}_cfg1 > gpurun_out/prof_${TAG}_cfg1.log 2>&1 || { tail -5 gpurun_out/prof_${TAG}_cfg1.log; exit 1; }
       python3 tools/make_traffic.py gpurun_out/prof_${TAG}_cfg1 IDX-D_20000000_150 profiles/${TAG}_cfg1_rocprofv3_summary.txt | tail -6 ;;
    2) PMC_SETS=min BENCH_ARGS="--config 2 --steps 3" bash tools/profile_gpu.sh ${TAG}_cfg2 > gpurun_out/prof_${TAG}_cfg2.log 2>&1 || { tail -5 gpurun_out/prof_${TAG}_cfg2.log; exit 1; }
       python3 tools/make_traffic.py gpurun_out/prof_${TAG}_cfg2 IDX-C_200000000_150 profiles/${TAG}_cfg2_rocprofv3_summary.txt | tail -6 ;;
    3) PMC_SETS=min BENCH_ARGS="--config 3 --steps 3" bash tools/profile_gpu.sh ${TAG}_cfg3 > gpurun_out/prof_${TAG}_cfg3.log 2>&1 || { tail -5 gpurun_out/prof_${TAG}_cfg3.log; exit 1; }
       python3 tools/make_traffic.py gpurun_out/prof_${TAG}_cfg3 IDX-D_200000000_150 profiles/${TAG}_cfg3_rocprofv3_summary.txt | tail -6 ;;
    4) PMC_SETS=min BENCH_ARGS="--config 4 --steps 1 --warmup 1" bash tools/profile_gpu.sh ${TAG}_cfg4 > gpurun_out/prof_${TAG}_cfg4.log 2>&1 || { tail -5 gpurun_out/prof_${TAG}_cfg4.log; exit 1; }
       python3 tools/make_traffic.py gpurun_out/prof_${TAG}_cfg4 config4_50000000x16_150 profiles/${TAG}_cfg4_rocprofv3_summary.txt --per-step 2 | tail -14 ;;
  esac
  echo "cfg$c done"
  cp gpurun_out/prof_${TAG}_cfg$c/summary.txt gpurun_out/${TAG}_cfg${c}_rocprofv3_summary.txt
  cp gpurun_out/prof_${TAG}_cfg$c/kernel_stats.csv gpurun_out/${TAG}_cfg${c}_kernel_stats.csv 2>/dev/null
  # the raw per-dispatch CSVs are tens of MB per configuration: gpurun copies back 64 MiB at most
  find gpurun_out/prof_${TAG}_cfg$c -name "*.csv" -size +512k -delete
done
cp profiles/hbm_traffic.json gpurun_out/hbm_traffic_${TAG}.json
