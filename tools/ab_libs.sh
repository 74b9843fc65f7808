for r in 1 2; do for lib in "$@"; do
  GFMATCH_LIB=/root/repo/genefuserust_amd/$lib python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-h2d --no-pack-sweep --no-stress --no-parity 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); r=j['roofline']
print('%-28s value %.3f G  pass %.4f ms  stages %s' % ('$lib', j['value']/1e9, r['kernel_ms_avg'], r['stage_ms']))"
done; done
