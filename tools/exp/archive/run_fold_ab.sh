#!/bin/bash
# round 4: fold / sums / arrival A/B on the GPU box (from the repo root); results under gpurun_out/fold_ab/
set -u
mkdir -p gpurun_out/fold_ab
OUT=$PWD/gpurun_out/fold_ab
./tools/exp/fold_ab 16 24 > $OUT/events.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof20 -o fold20 -- /root/repo/tools/exp/fold_ab 20 > $OUT/prof20.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/prof16 -o fold16 -- /root/repo/tools/exp/fold_ab 16 > $OUT/prof16.log 2>&1
find $OUT -name "*kernel_stats.csv" | head
