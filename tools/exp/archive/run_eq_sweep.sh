#!/bin/bash
mkdir -p gpurun_out/nt_sweep
for WG in 1 2 4; do for NB in 256 512 1024; do
  v=$(ZG_EQ_WG=$WG ZG_EQ_BLOCKS=$NB python tools/bench_eq.py --v 20 --reps 400 2>/dev/null | tail -1)
  echo "eq v=20 wg=$WG blocks=$NB: $v"
done; done | tee gpurun_out/nt_sweep/eq.txt
