#!/bin/bash
# Timing-only builds of the fused seed+verify kernel (ablations give wrong results by
# construction; occupancy variants are exact).  build here, run on the GPU box.
#   GF_ABLATE_SV: 6 = staging + conversion only, 5 = + the windows and seeds of every read (no look-ups),
#   noinl = the whole kernel without the inline filter pass (-DGF_SV_NO_INLINE_FILTER: exact, the
#   background reads go to gf_k_probe_filter instead)
#   GF_SVS_WAVES_PER_SIMD: register budget of the kernel
# (GF_ABLATE_SV 1 / 3 / 4 stop after the windows / seed filter / seed probes but leave every read
#  undecided: their times include 64 bytes of list entry per read.)
set -e
REPO=$(cd $(dirname $0)/.. && pwd)
VARIANTS="sv6:-DGF_ABLATE_SV=6 sv5:-DGF_ABLATE_SV=5 sv3:-DGF_ABLATE_SV=3 sv4:-DGF_ABLATE_SV=4 noinl:-DGF_SV_NO_INLINE_FILTER"
if [ "$1" = build ]; then
  for v in $VARIANTS; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 ${v#*:} -Wno-unused-function -shared -L/opt/rocm/lib -lrccl \
      -o $REPO/genefuserust_amd/libgfmatch_${v%%:*}.so $REPO/genefuserust_amd/csrc/gfmatch.hip &
  done
  wait
else
  for v in full $VARIANTS; do
    name=${v%%:*}
    lib=$REPO/genefuserust_amd/libgfmatch_$name.so; [ $name = full ] && lib=$REPO/genefuserust_amd/libgfmatch.so
    GFMATCH_LIB=$lib python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-parity --no-h2d --no-pack-sweep --no-stress "${@:2}" 2>/dev/null | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['roofline']['kernel_ms_avg'], d['roofline']['stage_ms'])"
  done
fi
