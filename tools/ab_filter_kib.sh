#!/bin/bash
# A/B on the GPU box: size of the L2-resident presence filter (GF_BLOOM_KIB) against seed+verify's time and
# its L2 hits / misses (one rocprofv3 --pmc pass each).  Usage: bash tools/ab_filter_kib.sh <tag> [bench args]
set -o pipefail
TAG=${1:-ab}; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/abkib_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for kib in ${KIBS:-1536 2048 2560 3072}; do
  export GF_BLOOM_KIB=$kib
  python3 $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-parity --no-h2d "$@" > $OUT/bench_$kib.json 2> $OUT/bench_$kib.err || { echo "bench $kib failed"; tail -3 $OUT/bench_$kib.err; exit 1; }
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_$kib -o pmc -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-h2d "$@" > $OUT/pmc_$kib.log 2>&1 || { echo "pmc $kib failed"; tail -3 $OUT/pmc_$kib.log; }
  python3 - $OUT $kib <<'PY'
import csv, glob, json, sys
out, kib = sys.argv[1], sys.argv[2]
j = json.load(open("%s/bench_%s.json" % (out, kib)))
acc = {}
for f in glob.glob("%s/pmc_%s/**/*counter_collection.csv" % (out, kib), recursive=True):
    for r in csv.DictReader(open(f)):
        if "gf_k_seedverify_stream<10, false>" in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
print("KIB %s: value %.3f G reads/s, stages %s, sv hits %.1f M misses %.1f M" % (
    kib, j["value"] / 1e9, j["roofline"]["stage_ms"], m.get("TCC_HIT_sum", 0) / 1e6, m.get("TCC_MISS_sum", 0) / 1e6))
PY
done
