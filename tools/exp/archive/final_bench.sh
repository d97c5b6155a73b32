#!/bin/bash
TAG=${1:-r3final}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_cpp_host.py tests/test_gpu_bench_multirank.py -q -x > gpurun_out/pytest_cpp_$TAG.txt 2>&1; tail -3 gpurun_out/pytest_cpp_$TAG.txt
timeout 1500 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; tail -c 600 gpurun_out/bench_$TAG.json; echo
timeout 600 tools/bench_sumcheck 20 5 > gpurun_out/bench_sumcheck_$TAG.json 2>&1; tail -c 2500 gpurun_out/bench_sumcheck_$TAG.json
