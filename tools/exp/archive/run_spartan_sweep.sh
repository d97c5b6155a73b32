#!/bin/bash
set -u
mkdir -p gpurun_out/spartan
OUT=$PWD/gpurun_out/spartan
cd /tmp && export TMPDIR=/tmp
for cfg in "512 2" "256 2" "1024 2" "512 1" "1024 1" "2048 1"; do
  set -- $cfg
  export ZG_SC_SPARTAN_BLOCKS=$1 ZG_SC_SPARTAN_WG=$2
  rocprofv3 --kernel-trace --stats -d $OUT/p_$1_$2 -o sc -- /root/repo/tools/bench_sumcheck > $OUT/log_$1_$2.txt 2>&1
done
