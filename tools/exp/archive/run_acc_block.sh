#!/bin/bash
mkdir -p gpurun_out/nt_sweep
for BL in ${BLS:-256 512}; do
  for NT in ${NTS:-131072 122880 114688 106496 98304}; do
    v=$(ZG_MSM_ACC_BLOCK=$BL ZG_MSM_INFLIGHT_CHUNKS=$NT python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extra </dev/null 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f MSM/s' % d['value'])")
    echo "block=$BL inflight=$NT: $v"
  done
done | tee gpurun_out/nt_sweep/acc_block2.txt
