#!/bin/bash
set -u
mkdir -p gpurun_out/cover
OUT=$PWD/gpurun_out/cover
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
for LOGN in 20 22; do
  rocprofv3 --kernel-trace -d $OUT/p$LOGN -o t -- python3 $ROOT/bench.py --logn $LOGN --steps 6 --warmup 2 --msms-per-step 8 --no-cpu-baseline --no-extra > $OUT/log$LOGN.txt 2>&1
  DB=$(find $OUT/p$LOGN -name "*results.db" | head -1)
  echo "== 2^$LOGN points, three streams"; grep -o '"value": [0-9.]*' $OUT/log$LOGN.txt | head -1
  python3 $ROOT/tools/exp/dbcover.py $DB accumulate
done 2>&1 | tee $OUT/cover.txt
