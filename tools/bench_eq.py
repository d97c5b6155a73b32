#!/usr/bin/env python3
"""eq table build (zg_fr_eq_table_dev) back to back on one stream: wall time per launch under the ZG_EQ_* environment the caller set.
    python tools/bench_eq.py [--v 20] [--reps 400]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zolt_amd import lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--v", type=int, default=20)
ap.add_argument("--reps", type=int, default=400)
ap.add_argument("--narrow", action="store_true", help="128-bit challenges as the reference's transcript produces them (stored [0, 0, lo, hi])")
args = ap.parse_args()
lib.init(0)
rng = np.random.default_rng(1)
r = lib.field_op(lib.FR, lib.OP_TO_MONT, rng.integers(0, 1 << 62, size=(args.v, 4), dtype=np.uint64))
if args.narrow:
    r[:, :2] = 0
    r[:, 3] &= np.uint64((1 << 61) - 1)
buf = lib.DeviceBuffer((1 << args.v) * 32)
for _ in range(20):
    lib.fr_eq_table_dev(r, buf.ptr)
lib.sync()
best = None
for _ in range(5):
    t0 = time.perf_counter()
    for _ in range(args.reps):
        lib.fr_eq_table_dev(r, buf.ptr)
    lib.sync()
    dt = (time.perf_counter() - t0) / args.reps
    best = dt if best is None else min(best, dt)
chk = buf.to_host()[:8].tolist()
print(json.dumps({"v": args.v, "narrow": args.narrow, "us_per_launch": best * 1e6, "GBps": (32 << args.v) / best / 1e9,
                  "env": {k: v for k, v in os.environ.items() if k.startswith("ZG_")}, "first_words": chk[:2]}))
