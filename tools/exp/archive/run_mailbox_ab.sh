#!/bin/bash
set -u
mkdir -p gpurun_out/mailbox_ab
OUT=$PWD/gpurun_out/mailbox_ab
ROOT=$PWD
./tools/exp/mailbox_ab > $OUT/host.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof -o mb -- $ROOT/tools/exp/mailbox_ab > $OUT/prof.log 2>&1
python3 $ROOT/tools/exp/dbstats.py $(find $OUT/prof -name "*results.db" | head -1) > $OUT/kernels.txt 2>&1
cat $OUT/host.txt $OUT/kernels.txt
