#!/bin/bash
set -u
mkdir -p gpurun_out/fold_ab2
OUT=$PWD/gpurun_out/fold_ab2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof20 -o fold20 -- /root/repo/tools/exp/fold_ab 20 > $OUT/prof20.log 2>&1
