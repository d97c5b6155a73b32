#!/bin/bash
mkdir -p gpurun_out/nt_sweep
for SF in 0 1; do
  for NT in 122880 114688 106496; do
    v=$(ZG_MSM_SLICE_SORT_FIRST=$SF ZG_MSM_INFLIGHT_CHUNKS=$NT python bench.py --logn 22 --steps 10 --warmup 2 --no-cpu-baseline --no-extra </dev/null 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f MSM/s' % d['value'])")
    echo "2^22 sort_first=$SF inflight=$NT: $v"
  done
done | tee gpurun_out/nt_sweep/e22.txt
for ST in 2 3 4; do
  v=$(python bench.py --logn 22 --steps 10 --warmup 2 --streams $ST --no-cpu-baseline --no-extra </dev/null 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f MSM/s' % d['value'])")
  echo "2^22 streams=$ST: $v"
done | tee -a gpurun_out/nt_sweep/e22.txt
