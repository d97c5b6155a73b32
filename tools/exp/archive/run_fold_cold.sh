#!/bin/bash
set -u
mkdir -p gpurun_out/fold_cold
OUT=$PWD/gpurun_out/fold_cold
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/prof -o cold -- python3 $ROOT/tools/exp/fold_cold.py 20 > $OUT/log.txt 2>&1
DB=$(find $OUT/prof -name "*results.db" | head -1)
python3 $ROOT/tools/exp/dbseq.py $DB sc_ > $OUT/seq.txt 2>&1
tail -60 $OUT/seq.txt
