#!/bin/bash
set -u
mkdir -p gpurun_out/r4c
OUT=$PWD/gpurun_out/r4c
python -m pytest tests/test_gpu_poly.py tests/test_gpu_handoff.py tests/test_gpu_prover_sites.py -x -q 2>&1 | tail -5 > $OUT/pytest_poly.txt
./tools/bench_sumcheck > $OUT/bench_sumcheck.json 2> $OUT/bench_sumcheck.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof -o sc -- /root/repo/tools/bench_sumcheck > $OUT/prof.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/prof20 -o fold20 -- /root/repo/tools/exp/fold_ab 20 > $OUT/prof20.log 2>&1
