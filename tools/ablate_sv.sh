#!/bin/bash
# Timing-only ablation builds of gf_k_seedverify (results are wrong by construction).
# Stage N = stop each read after: 1 stream loads, 2 clean-window bits, 3 presence
# filter, 4 seed probes, 5 verification (full = normal).  build here, run on the GPU box.
set -e
REPO=$(cd $(dirname $0)/.. && pwd)
if [ "$1" = build ]; then
  for n in 1 2 3 4 5; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DGF_ABLATE_SV=$n -shared \
      -o $REPO/genefuserust_amd/libgfmatch_sv$n.so $REPO/genefuserust_amd/csrc/gfmatch.hip
  done
else
  for n in 1 2 3 4 5 full; do
    lib=$REPO/genefuserust_amd/libgfmatch_sv$n.so; [ $n = full ] && lib=$REPO/genefuserust_amd/libgfmatch.so
    echo "== stage $n"
    GFMATCH_LIB=$lib bash $REPO/tools/trace_kernels.sh "${@:2}" | grep -E "seedverify"
  done
fi
