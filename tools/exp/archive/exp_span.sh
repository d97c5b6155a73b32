#!/bin/bash
# point slices (ZG_MSM_TABLE_SPAN_MB) on and off: whole-MSM times at 2^22 (and 2^24 when asked), serial and pipelined
mkdir -p gpurun_out
[ -n "$SKIP_TESTS" ] || timeout 900 python -m pytest tests/test_gpu_msm.py -q -x -k "point_slices or host_scalar_path_sliced or interleaves" 2>&1 | tail -3
for n in ${SIZES:-22}; do
  for mb in ${SPANS:-0 1024 512 2048}; do
    ZG_MSM_TABLE_SPAN_MB=$mb timeout 600 python tools/bench_tail.py --logn $n --reps 24 --streams 3 --tag "span_mb=$mb" 2>&1 | tail -1
  done
done > gpurun_out/exp_span.jsonl
cut -c1-420 gpurun_out/exp_span.jsonl
