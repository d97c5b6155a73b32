#!/bin/bash
for cfg in "20 1" "22 1" "22 0" "20 1" "22 1" "22 0"; do
  set -- $cfg
  ZG_MSM_SLICE_LOCAL_REFS=$2 timeout 600 python bench.py --logn $1 --steps 24 --warmup 4 --msms-per-step 8 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); e=d['extra']['kernel_ms_per_msm_alone']; print(json.dumps({'logn': $1, 'local_refs': $2, 'value': round(d['value'],1), 'sort_ms_per_msm': round(e['msm_sort'],3), 'acc_ms_per_msm': round(e['msm_accumulate'],3)}))"
done | tee gpurun_out/exp_local4.jsonl
timeout 900 python -m pytest tests/test_gpu_msm.py -q -x 2>&1 | tail -2
