#!/bin/bash
mkdir -p gpurun_out/nt_sweep
for NT in 131072 122880 114688 106496 114688; do
  v=$(ZG_MSM_INFLIGHT_CHUNKS=$NT python bench.py --precompute 1 --steps 12 --warmup 3 --no-cpu-baseline --no-extra </dev/null 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f MSM/s' % d['value'], d['config'].get('window_bits'))")
  echo "table-less inflight=$NT: $v"
done | tee gpurun_out/nt_sweep/noprecomp.txt
for L in 3 4 6 8; do
  v=$(ZG_MSM_LANES=$L python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extra </dev/null 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f MSM/s' % d['value'])")
  echo "lanes=$L: $v"
done | tee gpurun_out/nt_sweep/lanes.txt
