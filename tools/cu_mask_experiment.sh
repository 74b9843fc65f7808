#!/bin/bash
# Which resource binds the mapping pass: the default bench with fewer CUs (gpurun, from the repo root).
#   bash tools/cu_mask_experiment.sh [extra bench args]
# ROC_GLOBAL_CU_MASK restricts the CUs every kernel of the process may use.  On MI355X 32 consecutive bits are one
# XCD: a mask that keeps CUs in every XCD halves the CUs and leaves the L2s; a mask of whole XCDs halves both.
# Results of r02: profiles/r02_e_cu_mask_experiment.txt.
H5=$(printf '5%.0s' $(seq 64)); H0F=$(printf '0F%.0s' $(seq 32)); HX=$(printf '00000000FFFFFFFF%.0s' $(seq 4))
for m in full 0x$H5 0x$H0F 0x$HX; do
  if [ $m = full ]; then unset ROC_GLOBAL_CU_MASK; else export ROC_GLOBAL_CU_MASK=$m; fi
  python3 bench.py --no-cpu-baseline --no-h2d --no-parity --steps 5 "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('mask', '$m'[:20], round(d['value']/1e9,3), 'G reads/s', d['roofline']['stage_ms'])"
done
