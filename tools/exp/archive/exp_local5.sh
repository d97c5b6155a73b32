#!/bin/bash
# slices of handles whose own plan sorts in one pass (2^23, 2^24 points): two-pass slice sort on slice-local references, on / off.
# bench.py checks every timed MSM against the closed form.
for cfg in "23 1" "23 0" "24 1" "24 0"; do
  set -- $cfg
  ZG_MSM_SLICE_LOCAL_REFS=$2 timeout 900 python bench.py --logn $1 --steps 6 --warmup 2 --msms-per-step 4 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); e=d['extra']['kernel_ms_per_msm_alone']; print(json.dumps({'logn': $1, 'local_refs': $2, 'value': round(d['value'],2), 'ms_per_msm': round(d['config']['ms_per_msm'],3), 'sort_ms_per_msm': round(e['msm_sort'],3), 'acc_ms_per_msm': round(e['msm_accumulate'],3), 'launches': d['roofline']['launches_per_msm']}))"
done | tee gpurun_out/exp_local5.jsonl
timeout 900 python -m pytest tests/test_gpu_msm.py -q -x 2>&1 | tail -2
