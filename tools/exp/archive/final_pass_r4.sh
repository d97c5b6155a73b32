#!/bin/bash
# round 4: the evidence pass (GPU suite tail, bench line, rocprofv3 recipe for MSM + sumcheck family, compiled-host benches)
set -u
TAG=${1:-r4final}
mkdir -p gpurun_out/$TAG
OUT=$PWD/gpurun_out/$TAG
python -m pytest tests -m gpu -q 2>&1 | tail -8 > $OUT/pytest_gpu_tail.txt
python bench.py > $OUT/bench.json 2> $OUT/bench.err
./tools/bench_sumcheck > $OUT/bench_sumcheck.json 2> $OUT/bench_sumcheck.err
./tools/bench_sumcheck 13 10 > $OUT/bench_sumcheck_v13.json 2>> $OUT/bench_sumcheck.err
./tools/stage3_round_split 20 > $OUT/stage3_round_split.txt 2>&1
bash tools/profile_r4.sh $TAG > $OUT/profile.log 2>&1
