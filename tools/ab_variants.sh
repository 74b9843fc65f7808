#!/bin/bash
# A/B timing of map variants / alternative builds on the GPU box:
#   bash tools/ab_variants.sh "<lib>:<variant> ..." [bench args]
REPO=$(cd $(dirname $0)/.. && pwd)
for lv in $1; do
  lib=${lv%%:*}; v=${lv##*:}
  GFMATCH_LIB=$REPO/genefuserust_amd/$lib python3 $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline --variant $v "${@:2}" 2>/dev/null | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', 'variant $v', round(d['roofline']['kernel_ms_avg'],3), d['roofline']['stage_ms'], d['parity'])"
done
