#!/bin/bash
mkdir -p gpurun_out
for lib in zolt_amd/libzolt_gpu.so build_ab/libzolt_gpu_rowahead2.so; do
for cfg in "20 0 0" "22 0 0" "22 1024 0"; do
  set -- $cfg
  ZOLT_GPU_LIB=$lib ZG_MSM_TABLE_SPAN_MB=$2 ZG_MSM_SLICE_SORT_FIRST=$3 timeout 600 python bench.py --logn $1 --steps 12 --warmup 3 --msms-per-step 8 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(json.dumps({'lib': '$lib'[-14:], 'logn': $1, 'span_mb': $2, 'sort_first': $3, 'value': round(d['value'],1), 'ms_per_msm': round(d['config']['ms_per_msm'],3), 'acc_alone': round(d['extra']['kernel_ms_per_msm_alone']['msm_accumulate'],3)}))"
done; done > gpurun_out/exp_span3.jsonl
cat gpurun_out/exp_span3.jsonl
