#!/bin/bash
# A/B matrix for the fixed-cost part of an MSM (run on the GPU box from the repo root): window size and chunk length at the
# per-rank sizes of the strong-scaling run, serial and pipelined. One JSON line per setting into gpurun_out/ab_tail.jsonl.
OUT=${1:-gpurun_out/ab_tail.jsonl}
: > $OUT
run() { # tag logn env...
  tag=$1; logn=$2; shift 2
  env "$@" python3 tools/bench_tail.py --logn $logn --tag "$tag" >> $OUT 2>> ${OUT%.jsonl}.err
}
for logn in 17 18 19; do
  run default $logn ZG_NOOP=1
  run c15 $logn ZG_MSM_WINDOW_BITS=15
  run c14 $logn ZG_MSM_WINDOW_BITS=14
  run e32 $logn ZG_MSM_CHUNK_ENTRIES=32
  run e64 $logn ZG_MSM_CHUNK_ENTRIES=64
  run c15e32 $logn ZG_MSM_WINDOW_BITS=15 ZG_MSM_CHUNK_ENTRIES=32
  run c14e32 $logn ZG_MSM_WINDOW_BITS=14 ZG_MSM_CHUNK_ENTRIES=32
done
run default 20 ZG_NOOP=1
run nofull 20 ZG_MSM_ALONE_FULL=0
run c16 20 ZG_MSM_WINDOW_BITS=16
for logn in 4 8 10 13 15 16; do run default $logn ZG_NOOP=1; done
run c13 16 ZG_MSM_WINDOW_BITS=13
run c14 16 ZG_MSM_WINDOW_BITS=14
run c13 15 ZG_MSM_WINDOW_BITS=13
run c12 15 ZG_MSM_WINDOW_BITS=12
python3 - <<'PY'
import json
for l in open("gpurun_out/ab_tail.jsonl"):
    d = json.loads(l)
    print(f"2^{d['logn']:2d} {d['tag']:8s} plan={d['plan']} sync={d['serial_sync_ms']:.3f} one_stream={d['one_stream_ms']:.3f} pipelined={d['pipelined_ms']:.3f} {d['kernel_us_alone']} {d['result_x0'][-6:]}")
PY
