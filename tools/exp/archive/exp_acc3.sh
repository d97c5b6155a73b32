#!/bin/bash
# three accumulate waves per SIMD (168 registers, 47 spilled) against two (200 registers): bench.py at 2^20
for cfg in "zolt_amd/libzolt_gpu.so 0" "build_ab/libzolt_gpu_acc3.so 196608" "build_ab/libzolt_gpu_acc3.so 184320" "build_ab/libzolt_gpu_acc3.so 0"; do
  set -- $cfg
  if [ $2 = 0 ]; then unset ZG_MSM_CHUNK_THREADS; else export ZG_MSM_CHUNK_THREADS=$2; fi
  ZOLT_GPU_LIB=$1 timeout 600 python bench.py --steps 12 --warmup 3 --msms-per-step 8 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(json.dumps({'lib': '$1'[-12:], 'nt': $2, 'value': round(d['value'],1), 'acc_alone': round(d['roofline']['avg_launch_ms'],4)}))"
done | tee gpurun_out/exp_acc3.jsonl
