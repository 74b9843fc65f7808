#!/bin/bash
# The stress rows at the headline's size (GPU box): profiles/<tag>_stress_*.json = bench lines of configs[1]'s workload on
# harder gene sets / read mixes.  bash tools/stress_rows.sh <tag>
TAG=${1:-r04}
COMMON="--steps 10 --warmup 2 --no-cpu-baseline --no-h2d --no-pack-sweep --no-stress"
python bench.py $COMMON --repeat-frac 0.1 > gpurun_out/${TAG}_stress_repeat10.json 2>/dev/null
python bench.py $COMMON --repeat-frac 0.3 > gpurun_out/${TAG}_stress_repeat30.json 2>/dev/null
python bench.py $COMMON --low-complexity 0.05 > gpurun_out/${TAG}_stress_lowcomplexity5.json 2>/dev/null
python bench.py $COMMON --shape IDX-T > gpurun_out/${TAG}_stress_idxt.json 2>/dev/null
python bench.py $COMMON --mix WGS > gpurun_out/${TAG}_stress_wgs.json 2>/dev/null
for f in repeat10 repeat30 lowcomplexity5 idxt wgs; do
  python - gpurun_out/${TAG}_stress_$f.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).readline()); r = j["roofline"]
print("%-40s %.3f G reads/s  pass %.3f ms  %s  parity %s hits %d" % (sys.argv[1].split("/")[-1], j["value"] / 1e9, r["kernel_ms_avg"], r["stage_ms"],
      j["parity"]["bit_exact"], j["config"]["hits_per_step"]))
PY
done
