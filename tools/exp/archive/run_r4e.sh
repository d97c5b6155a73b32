#!/bin/bash
# round 4, pass e: full GPU suite, CPU baseline at 2^22 in full, the bench line, crossover of the one-shot MSM
set -u
mkdir -p gpurun_out/r4e
OUT=$PWD/gpurun_out/r4e
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $OUT/pytest_gpu.txt
python bench.py --cpu-baseline-2e22 > $OUT/cpu_baseline_2e22.json 2> $OUT/cpu_baseline_2e22.err
cp profiles/cpu_baseline_2e22_full.json $OUT/ 2>/dev/null
python bench.py > $OUT/bench.json 2> $OUT/bench.err
