#!/bin/bash
# upper bound of what hiding digits + sort completely would buy the pipelined rate (the same scalars every call, so a stale sorted list is
# the right one): bench_tail at 2^20, three streams
for lib in zolt_amd/libzolt_gpu.so build_ab/libzolt_gpu_skipsort.so zolt_amd/libzolt_gpu.so build_ab/libzolt_gpu_skipsort.so; do
  ZOLT_GPU_LIB=$lib timeout 300 python tools/bench_tail.py --logn 20 --reps 300 --streams 3 --tag "$lib" 2>&1 | tail -1 | cut -c1-330
done | tee gpurun_out/exp_skipsort.jsonl
