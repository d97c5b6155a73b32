#!/bin/bash
# round 4, pass b: hand-off stress test, poly/sumcheck GPU tests, bench_sumcheck with per-kernel durations
set -u
mkdir -p gpurun_out/r4b
OUT=$PWD/gpurun_out/r4b
python -m pytest tests/test_gpu_handoff.py -x -q 2>&1 | tail -5 > $OUT/pytest_handoff.txt
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $OUT/pytest_gpu.txt
./tools/bench_sumcheck > $OUT/bench_sumcheck.json 2> $OUT/bench_sumcheck.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof -o sc -- /root/repo/tools/bench_sumcheck > $OUT/prof.log 2>&1
