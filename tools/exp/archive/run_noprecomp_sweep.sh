#!/bin/bash
# round 4: table-less MSM (precompute_levels = 1) by window size, 2^20 and 2^16 points
set -u
mkdir -p gpurun_out/noprecomp
OUT=$PWD/gpurun_out/noprecomp
for logn in 20 16; do
for c in 10 11 12 13 14 15 16; do
  python bench.py --logn $logn --steps 4 --warmup 1 --msms-per-step 8 --precompute 1 --window-bits $c --no-cpu-baseline --no-extra > $OUT/l${logn}_c$c.json 2> $OUT/l${logn}_c$c.err
  python bench.py --logn $logn --steps 4 --warmup 1 --msms-per-step 8 --precompute 1 --window-bits $c --no-cpu-baseline --no-extra --streams 1 > $OUT/l${logn}_c${c}_serial.json 2>> $OUT/l${logn}_c$c.err
done
done
