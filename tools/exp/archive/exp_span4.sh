#!/bin/bash
mkdir -p gpurun_out
for mb in 0 1024 1400 1900 0 1024; do
  ZG_MSM_TABLE_SPAN_MB=$mb timeout 600 python bench.py --logn 22 --steps 12 --warmup 3 --msms-per-step 8 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(json.dumps({'logn': 22, 'span_mb': $mb, 'value': round(d['value'],1), 'ms_per_msm': round(d['config']['ms_per_msm'],3), 'acc_alone': round(d['extra']['kernel_ms_per_msm_alone']['msm_accumulate'],3)}))"
done > gpurun_out/exp_span4.jsonl
cat gpurun_out/exp_span4.jsonl
