#!/bin/bash
# Timing-only ablation builds of the short-read mapping kernel (results are wrong by
# construction; only kernel_ms matters).  Stage N = stop each read after stage N:
#   1 staging  2 key extraction  3 seed probes  4 verification + bound check  (full = normal)
# Build here (CPU container), run on the GPU box:  bash tools/ablate.sh build | run
set -e
REPO=$(cd $(dirname $0)/.. && pwd)
if [ "$1" = build ]; then
  for n in 1 2 3 4; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DGF_ABLATE=$n -shared \
      -o $REPO/genefuserust_amd/libgfmatch_abl$n.so $REPO/genefuserust_amd/csrc/gfmatch.hip
  done
else
  for n in 1 2 3 4 full; do
    lib=$REPO/genefuserust_amd/libgfmatch_abl$n.so; [ $n = full ] && lib=$REPO/genefuserust_amd/libgfmatch.so
    GFMATCH_LIB=$lib python3 $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-parity "${@:2}" | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stage $n kernel_ms %.3f' % d['roofline']['kernel_ms_avg'])"
  done
fi
