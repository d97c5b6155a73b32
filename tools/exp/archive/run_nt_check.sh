#!/bin/bash
# the in-flight chunk count (ZG_MSM_INFLIGHT_CHUNKS) at the other sizes: 122880 = the old 15/16, 114688 = 7/8
mkdir -p gpurun_out/nt_sweep
for LOGN in 17 18 19 22; do
  for NT in 122880 114688; do
    v=$(ZG_MSM_INFLIGHT_CHUNKS=$NT python bench.py --logn $LOGN --steps 12 --warmup 3 --no-cpu-baseline --no-extra </dev/null 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f MSM/s' % d['value'])")
    echo "logn=$LOGN inflight=$NT: $v"
  done
done | tee gpurun_out/nt_sweep/other_sizes.txt
