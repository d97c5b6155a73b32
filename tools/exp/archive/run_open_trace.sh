#!/bin/bash
set -u
mkdir -p gpurun_out/open_trace
OUT=$PWD/gpurun_out/open_trace
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/prof -o open -- python3 $ROOT/tools/exp/open_trace.py > $OUT/log.txt 2>&1
tail -2 $OUT/log.txt
DB=$(find $OUT/prof -name "*results.db" | head -1)
python3 $ROOT/tools/exp/dbtimeline.py $DB > $OUT/timeline.txt 2>&1
cat $OUT/timeline.txt
