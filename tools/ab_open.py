#!/usr/bin/env python3
"""A/B helper: HyperKZG.open of 2^16 and 2^20 evaluations on a 2^20-point SRS (bench.py's extra), one JSON line.
Run with ZOLT_GPU_LIB=<other build> to compare two builds of the library on the same box."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import raw_scalars
from zolt_amd import api, lib
lib.init(0)
n = 1 << 20
g = api.generator()
ks = np.zeros((n, 4), dtype=np.uint64); ks[:, 0] = np.arange(1, n + 1, dtype=np.uint64)
xy, inf = lib.g1_scalar_mul_batch(np.repeat(g[None, :], n, axis=0), np.zeros(n, dtype=np.uint8), lib.field_op(lib.FR, lib.OP_TO_MONT, ks))
params = api.HyperKZG.SetupParams(xy, inf)
res = {"lib": os.path.basename(os.path.dirname(os.path.dirname(lib.LIB_PATH)))}
for vv in (12, 16, 20):
    ev = lib.field_op(lib.FR, lib.OP_TO_MONT, raw_scalars(0x4F50454E + vv, 0, 1 << vv))
    pt = lib.field_op(lib.FR, lib.OP_TO_MONT, raw_scalars(0x50543030 + vv, 0, vv))
    for _ in range(2):
        api.HyperKZG.open(params, ev, pt, np.zeros(4, dtype=np.uint64))
    ts = []
    for _ in range(7):
        t0 = time.perf_counter()
        api.HyperKZG.open(params, ev, pt, np.zeros(4, dtype=np.uint64))
        ts.append((time.perf_counter() - t0) * 1e3)
    res[f"v{vv}_ms_min"] = min(ts)
    res[f"v{vv}_ms_med"] = sorted(ts)[3]
print(json.dumps(res))
