#!/usr/bin/env python3
"""HyperKZG.open of 2^20 resident evaluations, a few calls, for a kernel trace (which launch sets overlap, where the call waits):
  rocprofv3 --kernel-trace -d out -o open -- python3 tools/exp/open_trace.py ; tools/exp/dbtimeline.py out/.../open_results.db"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import raw_scalars
from zolt_amd import api, lib
lib.init(0)
n = 1 << 20
g = api.generator()
ks = np.zeros((n, 4), dtype=np.uint64); ks[:, 0] = np.arange(1, n + 1, dtype=np.uint64)
xy, inf = lib.g1_scalar_mul_batch(np.repeat(g[None, :], n, axis=0), np.zeros(n, dtype=np.uint8), lib.field_op(lib.FR, lib.OP_TO_MONT, ks))
srs = lib.Bases.upload(xy, inf)
v = 20
ev = lib.field_op(lib.FR, lib.OP_TO_MONT, raw_scalars(0x4F50454E + v, 0, 1 << v))
pt = lib.field_op(lib.FR, lib.OP_TO_MONT, raw_scalars(0x50543030 + v, 0, v))
d_ev = lib.DeviceBuffer.from_host(ev)
ts = []
for i in range(6):
    lib.sync()
    time.sleep(0.003)
    t0 = time.perf_counter()
    lib.hyperkzg_open_dev(srs, d_ev.ptr, 1 << v, pt, np.zeros(4, dtype=np.uint64))
    ts.append((time.perf_counter() - t0) * 1e3)
print("open ms:", ["%.3f" % t for t in ts])
