#!/bin/bash
set -u
mkdir -p gpurun_out/small
OUT=$PWD/gpurun_out/small
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
for N in 16 1024; do
  python3 $ROOT/tools/bench_small_msm.py $N 2>/dev/null | tail -1
  rocprofv3 --kernel-trace -d $OUT/p$N -o t -- python3 $ROOT/tools/bench_small_msm.py $N > $OUT/log$N.txt 2>&1
  DB=$(find $OUT/p$N -name "*results.db" | head -1)
  echo "== $N points: the last MSM's kernels"; python3 $ROOT/tools/exp/dbtimeline.py $DB | tail -20
done 2>&1 | tee $OUT/small.txt
