#!/bin/bash
set -u
mkdir -p gpurun_out/tail
OUT=$PWD/gpurun_out/tail
cd /tmp && export TMPDIR=/tmp
for tm in 64 128 512 4096; do
  export ZG_SC_TAIL_MAX=$tm
  rocprofv3 --kernel-trace --stats -d $OUT/p_$tm -o sc -- /root/repo/tools/bench_sumcheck > $OUT/log_$tm.txt 2>&1
done
