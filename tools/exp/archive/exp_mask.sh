#!/bin/bash
mkdir -p gpurun_out
for n in 20 22; do
  for v in "" mask0xFFFFFF mask0x3FFFFF mask0xFFFFF; do
    if [ -z "$v" ]; then L=zolt_amd/libzolt_gpu.so; else L=build_ab/libzolt_gpu_$v.so; fi
    ZOLT_GPU_LIB=$L timeout 300 python tools/bench_tail.py --logn $n --reps 40 --tag "base$v" 2>&1 | tail -1
  done
done > gpurun_out/exp_mask.jsonl
cat gpurun_out/exp_mask.jsonl | cut -c1-600
