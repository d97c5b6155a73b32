#!/bin/bash
set -u
mkdir -p gpurun_out/noprecomp_prof
OUT=$PWD/gpurun_out/noprecomp_prof
cd /tmp && export TMPDIR=/tmp
for cfg in "20 15" "16 13"; do
  set -- $cfg
  rocprofv3 --kernel-trace --stats -d $OUT/p_$1_$2 -o m -- python3 /root/repo/bench.py --logn $1 --steps 3 --warmup 1 --msms-per-step 4 --precompute 1 --window-bits $2 --no-cpu-baseline --no-extra --streams 1 > $OUT/log_$1_$2.txt 2>&1
done
