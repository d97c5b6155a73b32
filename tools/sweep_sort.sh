#!/bin/bash
# two-pass sort tuning: scalars per partition block x fine bits; prints serial per-phase times and pipelined MSM/s
for SPAN in ${1:-512 1024 2048}; do
  for FB in ${2:-5 6 7}; do
    v=$(ZG_MSM_TWO_PASS_SPAN=$SPAN ZG_MSM_FINE_BITS=$FB timeout 120 python bench.py --logn ${3:-20} --steps 40 --warmup 6 --no-cpu-baseline --no-extra --full-line --streams 3 </dev/null 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f MSM/s' % d['value'], {k: round(v,3) for k,v in d['extra']['kernel_ms_per_msm_alone'].items()})")
    echo "span=$SPAN fb=$FB: $v"
  done
done
