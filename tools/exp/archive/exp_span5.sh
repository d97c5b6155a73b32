#!/bin/bash
# slice span at 2^22 (repeats, longer runs) and at 2^24
for rep in 1 2; do for mb in 0 1024 1400; do
  ZG_MSM_TABLE_SPAN_MB=$mb timeout 600 python bench.py --logn 22 --steps 24 --warmup 4 --msms-per-step 8 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(json.dumps({'logn': 22, 'span_mb': $mb, 'value': round(d['value'],1), 'launches': d['roofline']['launches_per_msm']}))"
done; done | tee gpurun_out/exp_span5.jsonl
for mb in 1024 1400 2048; do
  ZG_MSM_TABLE_SPAN_MB=$mb timeout 600 python tools/bench_tail.py --logn 24 --reps 12 --streams 3 --tag "span_mb=$mb" 2>&1 | tail -1 | cut -c1-200
done | tee -a gpurun_out/exp_span5.jsonl
