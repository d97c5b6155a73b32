#!/bin/bash
mkdir -p gpurun_out/nt_sweep
for Q in 4 2 8 4; do
  v=$(ZG_MSM_COMBINE_PER_QUAD=$Q python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extra </dev/null 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f MSM/s' % d['value'], {k: round(v,3) for k,v in d['extra']['kernel_ms_per_msm_alone'].items()})")
  echo "combine_per_quad=$Q: $v"
done | tee gpurun_out/nt_sweep/combine.txt
