#!/usr/bin/env python3
"""One MSM size, serial and pipelined, under whatever ZG_MSM_* environment the caller set — the A/B driver for the fixed-cost part
of an MSM (sort, bucket combine, reduction, final): what a rank sees at 2^17 points per GPU in the 8-GPU strong-scaling run, and
what a short commitment costs.

    python tools/bench_tail.py --logn 17 [--streams 3] [--reps 200] [--tag name]
Prints one JSON line: serial ms per MSM (one stream, nothing else in flight), pipelined ms per MSM (round-robin on --streams
streams), and the per-kernel-group HIP-event times of the serial run."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zolt_amd import api, lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--logn", type=int, default=17)
ap.add_argument("--streams", type=int, default=3)
ap.add_argument("--reps", type=int, default=200)
ap.add_argument("--tag", default="")
args = ap.parse_args()

lib.init(0)
dev = torch.device("cuda", 0)
n = 1 << args.logn
g = api.generator()
ks = np.zeros((n, 4), dtype=np.uint64)
ks[:, 0] = np.arange(1, n + 1, dtype=np.uint64)
bases_xy, _ = lib.g1_scalar_mul_batch(np.repeat(g[None, :], n, axis=0), np.zeros(n, dtype=np.uint8), lib.field_op(lib.FR, lib.OP_TO_MONT, ks))
rng = np.random.default_rng(7)
raw = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
sc = torch.from_numpy(lib.field_op(lib.FR, lib.OP_TO_MONT, raw).view(np.int64)).to(dev)
streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, args.streams))]
d_b = torch.from_numpy(bases_xy.view(np.int64)).to(dev)
b = lib.Bases.upload_dev(d_b.data_ptr(), 0, n, stream=streams[0].cuda_stream)
outs = torch.zeros((len(streams), 9), dtype=torch.int64, device=dev)
torch.cuda.synchronize()


def run(reps, nst):
    for i in range(reps):
        j = i % nst
        b.msm_dev_async(sc.data_ptr(), n, outs[j].data_ptr(), outs[j, 8:].data_ptr(), stream=streams[j].cuda_stream)
    torch.cuda.synchronize()


res = {"tag": args.tag, "logn": args.logn, "plan": b.plan(), "env": {k: v for k, v in os.environ.items() if k.startswith("ZG_")}}
run(6, 1)
first = outs[0].cpu().numpy().copy()
# serial: each MSM waits for the previous one (host-synchronised, what one synchronous call costs without the D2H)
t0 = time.perf_counter()
for _ in range(30):
    run(1, 1)
res["serial_sync_ms"] = (time.perf_counter() - t0) / 30 * 1e3
t0 = time.perf_counter()
run(args.reps, 1)
res["one_stream_ms"] = (time.perf_counter() - t0) / args.reps * 1e3
run(6, len(streams))
t0 = time.perf_counter()
run(args.reps, len(streams))
res["pipelined_ms"] = (time.perf_counter() - t0) / args.reps * 1e3
lib.profile_begin(8 * 8 + 8)
for _ in range(8):
    run(1, 1)
prof = lib.profile_end()
res["kernel_us_alone"] = {k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in prof.items() if v[1]}
assert np.array_equal(outs[0].cpu().numpy(), first) and np.array_equal(outs[-1].cpu().numpy(), first)
res["result_x0"] = hex(int(first.view(np.uint64)[0]))  # the same for every setting: A/B runs compare it
print(json.dumps(res))
