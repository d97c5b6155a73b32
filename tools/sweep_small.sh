#!/bin/bash
# small-n tuning sweep: scalars per sort block x window bits, pipelined on 3 streams
for LOGN in ${1:-17 18}; do
  for C in ${2:-15 16}; do
    for SPAN in ${3:-512 1024 2048}; do
      v=$(ZG_MSM_WINDOW_BITS=$C ZG_MSM_SORT_SPAN=$SPAN timeout 120 python bench.py --logn $LOGN --steps 48 --warmup 8 --no-cpu-baseline --no-extra --full-line --streams 3 </dev/null 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f MSM/s %.4f ms' % (d['value'], d['ms_per_step']), {k: round(v,3) for k,v in d['extra']['kernel_ms_per_msm'].items()})")
      echo "logn=$LOGN c=$C span=$SPAN: $v"
    done
  done
done
