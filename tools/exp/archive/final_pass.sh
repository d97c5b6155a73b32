#!/bin/bash
# the round's evidence pass on the GPU box: full GPU suite, the default bench line, the rocprofv3 recipe
TAG=${1:-r3final}
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -q -x -m gpu > gpurun_out/pytest_gpu_$TAG.txt 2>&1; tail -3 gpurun_out/pytest_gpu_$TAG.txt
timeout 1500 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; tail -c 1500 gpurun_out/bench_$TAG.json
timeout 2400 bash tools/profile_r3.sh $TAG > gpurun_out/profile_$TAG.log 2>&1; tail -3 gpurun_out/profile_$TAG.log
ls gpurun_out/prof_$TAG gpurun_out/prof_${TAG}_2e22
