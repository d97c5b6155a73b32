#!/bin/bash
mkdir -p gpurun_out
for n in 4 10 17 20; do
  timeout 600 python tools/bench_tail.py --logn $n --reps 200 --streams 4 --tag "r3y" 2>&1 | tail -1 | cut -c1-420
done > gpurun_out/exp_tail.jsonl
python tools/bench_host_path.py > gpurun_out/exp_host_path.txt 2>&1
cat gpurun_out/exp_tail.jsonl; tail -12 gpurun_out/exp_host_path.txt
timeout 3000 python -m pytest tests -q -x -m gpu > gpurun_out/pytest_gpu_r3y.txt 2>&1; tail -3 gpurun_out/pytest_gpu_r3y.txt
