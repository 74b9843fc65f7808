#!/bin/bash
# A/B of one environment switch on the headline pass (GPU box): bash tools/ab_env.sh NAME=VALUE [rounds] [extra bench args]
# alternates `NAME=VALUE python bench.py ...` and the default, prints seed+verify's and the pass's kernel time of each run.
SW=$1; ROUNDS=${2:-3}; shift; shift
for i in $(seq 1 $ROUNDS); do
  for mode in A B; do
    if [ $mode = A ]; then pre="env $SW"; tag="$SW"; else pre="env"; tag="default"; fi
    $pre python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-h2d --no-pack-sweep --no-stress --no-parity "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); r=j['roofline']
print('%-28s value %.3f G  pass %.4f ms  stages %s' % ('$tag', j['value']/1e9, r['kernel_ms_avg'], r['stage_ms']))"
  done
done
