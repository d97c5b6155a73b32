#!/bin/bash
# sweep workspace lanes / streams per problem size (throughput of pipelined MSMs); prints MSM/s
# usage: sweep_lanes.sh "17 18" "3 6 8" [GPU_MAX_HW_QUEUES]
for LOGN in ${1:-17 18 19 20}; do
  for LS in ${2:-3 4 6 8}; do
    v=$(GPU_MAX_HW_QUEUES=${3:-4} ZG_MSM_LANES=$LS timeout 120 python bench.py --logn $LOGN --steps 48 --warmup 8 --no-cpu-baseline --no-extra --streams $LS </dev/null 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f MSM/s %.4f ms' % (d['value'], d['ms_per_step']))")
    echo "logn=$LOGN lanes=streams=$LS hwq=${3:-4}: $v"
  done
done
