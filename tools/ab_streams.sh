#!/bin/bash
OUT=gpurun_out/ab_streams.jsonl
: > $OUT
run() { tag=$1; logn=$2; st=$3; shift 3; env "$@" python3 tools/bench_tail.py --logn $logn --streams $st --tag "$tag" >> $OUT 2>> gpurun_out/ab_streams.err; }
for logn in 17 18 19 20; do
  for st in 3 4 6 8; do run s$st $logn $st ZG_MSM_LANES=8; done
done
run s6q2 17 6 ZG_MSM_LANES=8 ZG_MSM_COMBINE_PER_QUAD=2
run s6q8 17 6 ZG_MSM_LANES=8 ZG_MSM_COMBINE_PER_QUAD=8
run s6c15 17 6 ZG_MSM_LANES=8 ZG_MSM_WINDOW_BITS=15
run s6e32 17 6 ZG_MSM_LANES=8 ZG_MSM_CHUNK_ENTRIES=32
run s6c15e32 17 6 ZG_MSM_LANES=8 ZG_MSM_WINDOW_BITS=15 ZG_MSM_CHUNK_ENTRIES=32
run s6c15 18 6 ZG_MSM_LANES=8 ZG_MSM_WINDOW_BITS=15
run s3q2 20 3 ZG_MSM_COMBINE_PER_QUAD=2
run s3q8 20 3 ZG_MSM_COMBINE_PER_QUAD=8
run s8 16 8 ZG_MSM_LANES=8
run s8 13 8 ZG_MSM_LANES=8
run s8 10 8 ZG_MSM_LANES=8
run s8 4 8 ZG_MSM_LANES=8
python3 - <<'PY'
import json
for l in open("gpurun_out/ab_streams.jsonl"):
    d = json.loads(l)
    print(f"2^{d['logn']:2d} {d['tag']:9s} plan={d['plan']} sync={d['serial_sync_ms']:.3f} one_stream={d['one_stream_ms']:.3f} pipelined={d['pipelined_ms']:.3f} {d['kernel_us_alone']} {d['result_x0'][-6:]}")
PY
