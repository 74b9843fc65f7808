#!/bin/bash
# (r01: a TA_* pass aborted inside rocprofv3 on this pool and was dropped without its log.  r02 asked again and
#  kept the log (profiles/r02_deep_ta_pass_abort.log): "Could not construct profile cfg failed with error code 38:
#  Request exceeds the capabilities of the hardware to collect" — three TA counters in one pass are more than the
#  TA block's slots; rocprofv3 turns that into a fatal check (SIGABRT) before the program runs a kernel.  It is a
#  profiler configuration error, not a fault of the profiled program.  TA counters go two per pass below.)
# Deeper PMC passes for the mapping kernel (instruction mix, TA/TCP stalls, TLB).
#   bash tools/profile_deep.sh <tag> [extra bench args]
set -o pipefail
TAG=${1:-deep}; shift
REPO=$(pwd)
KEEP=$REPO/gpurun_out/prof_$TAG   # summaries and logs (gpurun copies back at most 64 MiB)
OUT=/tmp/prof_$TAG                # raw counter CSVs stay on the box
mkdir -p $OUT $KEEP
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 3 --warmup 1 --profile-mode $*"
rocprofv3 --list-avail > $OUT/list_avail.txt 2>&1 || true
i=0
for pmc in \
 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY" \
 "SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM" \
 "SQ_BUSY_CYCLES SQ_LEVEL_WAVES SQ_INSTS SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_IFETCH" \
 "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
 "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
 "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
 "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" \
 "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" ; do
  for c in $pmc; do grep -q "${c%_sum}" $OUT/list_avail.txt || echo "note: $c is not in rocprofv3 --list-avail on this box"; done
  i=$((i+1))
  echo "pass $i: $pmc"
  timeout -k 10 150 rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc_$i -o pmc -- $BENCH > $OUT/pmc_$i.log 2>&1 || { echo "pmc pass $i ($pmc) failed"; tail -3 $OUT/pmc_$i.log; }
done
cd $REPO
python3 tools/summarize_profile.py $OUT > $KEEP/summary.txt 2>&1
cp $OUT/*.log $OUT/list_avail.txt $KEEP/ 2>/dev/null
cat $KEEP/summary.txt
