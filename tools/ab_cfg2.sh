#!/bin/bash
# IDX-C (BASELINE configs[2]) at 40 M reads per pass: bash tools/ab_cfg2.sh "NAME=VALUE ..." ["NAME=VALUE ..." ...]
for sw in "$@"; do
  env $sw python bench.py --config 2 --pairs 20000000 --steps 5 --warmup 1 --no-cpu-baseline --no-h2d --no-pack-sweep --no-stress 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); r=j['roofline']
print('%-50s value %.3f G  pass %.4f ms  stages %s parity %s' % ('$sw', j['value']/1e9, r['kernel_ms_avg'], r['stage_ms'], j['parity']['bit_exact']))"
done
