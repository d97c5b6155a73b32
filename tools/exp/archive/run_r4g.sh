#!/bin/bash
set -u
mkdir -p gpurun_out/r4g
OUT=$PWD/gpurun_out/r4g
python -m pytest tests/test_gpu_product_sumcheck.py tests/test_gpu_stage3.py tests/test_gpu_prover_sites.py tests/test_gpu_cpp_host.py tests/test_gpu_psc_grid.py tests/test_gpu_narrow_challenges.py -x -q 2>&1 | tail -5 > $OUT/pytest_psc.txt
./tools/bench_sumcheck > $OUT/bench_sumcheck.json 2> $OUT/bench_sumcheck.err
ZG_PSC_SPREAD_MAX_PAIRS=0 ./tools/bench_sumcheck > $OUT/bench_sumcheck_nospread.json 2>> $OUT/bench_sumcheck.err
ZG_PSC_SPREAD_MAX_PAIRS=4096 ./tools/bench_sumcheck > $OUT/bench_sumcheck_spread4096.json 2>> $OUT/bench_sumcheck.err
ZG_PSC_SPREAD_MAX_PAIRS=256 ./tools/bench_sumcheck > $OUT/bench_sumcheck_spread256.json 2>> $OUT/bench_sumcheck.err
./tools/bench_sumcheck 13 10 > $OUT/bench_sumcheck_v13.json 2>> $OUT/bench_sumcheck.err
ZG_PSC_SPREAD_MAX_PAIRS=0 ./tools/bench_sumcheck 13 10 > $OUT/bench_sumcheck_v13_nospread.json 2>> $OUT/bench_sumcheck.err
