#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root:
#   bash tools/profile_gpu.sh <tag>
# 1) rocprofv3 --kernel-trace --stats of the default bench command
# 2) separate --pmc passes for HBM traffic (FETCH_SIZE / WRITE_SIZE cannot share a pass)
# Summaries land in gpurun_out/prof_<tag>/ ; copy what is to be judged into profiles/.
set -o pipefail
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# BENCH_ARGS: extra bench.py arguments (e.g. "--config 2 --pairs 10000000" for IDX-C at 20 M reads);
# PMC_SETS: "all" (default), "traffic" (FETCH/WRITE/TCC only) or "min" (three passes: bytes and L2 hits / misses)
BENCH="python3 $REPO/bench.py --steps 5 --warmup 1 --profile-mode ${BENCH_ARGS:-}"
PMC_SETS=${PMC_SETS:-all}
# which build these counters describe: hashes of the sources and of the library on THIS box (tools/make_traffic.py adds the commit)
python3 - "$OUT" "${BENCH_ARGS:-}" <<'PY'
import json, sys
sys.path.insert(0, ".")
sys.path.insert(0, __import__("os").environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
json.dump({"kernel_src_sha": bench.kernel_source_sha(), "lib_sha": bench.library_sha(), "bench_args": sys.argv[2]},
          open(sys.argv[1] + "/build_stamp.json", "w"))
PY
echo "== kernel trace ==" 
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/trace_bench.log 2>&1 || { echo trace failed; tail -20 $OUT/trace_bench.log; exit 1; }
# (r03: TCC_BUBBLE = the 128-byte read requests FETCH_SIZE's gfx950 formula relies on; the _DRAM_32B counter
#  tallies a 64-byte request as 2 and a 128-byte one as 4, i.e. bytes / 32 whatever the request size)
SETS=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum"
      "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_32B_sum" "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum"
      "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum")
if [ "$PMC_SETS" = "min" ]; then   # what tools/make_traffic.py needs, nothing else (the 200 M-read configuration)
  SETS=("WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_32B_sum")
fi
if [ "$PMC_SETS" = "all" ]; then
  SETS+=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE")
fi
for pmc in "${SETS[@]}"; do
  name=$(echo $pmc | tr ' ' '_' | cut -c1-40)
  echo "== pmc $pmc =="
  rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc_$name -o pmc -- $BENCH > $OUT/pmc_${name}.log 2>&1 || { echo "pmc $pmc failed"; tail -5 $OUT/pmc_${name}.log; }
done
cd $REPO
python3 tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \; 2>/dev/null
cat $OUT/summary.txt
