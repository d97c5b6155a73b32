#!/bin/bash
# the torchrun code path of bench.py (partial MSM, RCCL all-gather, device combine) on ONE GPU at the per-rank sizes of the strong-scaling
# run, over stream counts and HW-queue counts: which setting should world > 1 use?
export ZOLT_BENCH_FORCE_SHARDED=1 WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1
port=29800
for logn in 17 19; do
  for q in 4 8; do
    for s in 3 4 6; do
      port=$((port+1))
      GPU_MAX_HW_QUEUES=$q MASTER_PORT=$port python3 bench.py --logn $logn --steps 10 --warmup 3 --msms-per-step 32 --no-cpu-baseline --no-extra --streams $s 2>/dev/null \
        | grep '^{' | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('logn=$logn hwq=$q streams=$s', round(d['value'],1), 'MSM/s', round(d['config']['ms_per_msm'],4), 'ms/MSM')"
    done
  done
done
