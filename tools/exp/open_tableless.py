#!/usr/bin/env python3
"""HyperKZG.open of 2^v resident evaluations over a key WITHOUT the table of multiples (zg_msm_config.expected_uses = 1), beside the key
with the table: python3 tools/exp/open_tableless.py [v] [uses]; under rocprofv3 --kernel-trace --stats it gives the per-kernel split."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from zolt_amd import lib, api
v = int(sys.argv[1]) if len(sys.argv) > 1 else 20
uses = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lib.init(0)
n = 1 << v
h, _, _ = lib.Bases.hyperkzg_setup(api.generator(), api.fr_from_int(0x12345678), n, want_points=False, expected_uses=uses)
print("plan (c, windows, levels):", h.plan(), "table bytes", h.table_bytes())
rng = np.random.default_rng(3)
ev = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); ev[:, 3] >>= 4
pt = rng.integers(0, 1 << 62, size=(v, 4), dtype=np.uint64); pt[:, 3] >>= 4
d = lib.DeviceBuffer.from_host(ev)
for rep in range(4):
    t0 = time.perf_counter()
    q, qi, f = lib.hyperkzg_open_dev(h, d.ptr, n, pt, np.zeros(4, dtype=np.uint64))
    print("open ms", (time.perf_counter() - t0) * 1e3)
sc = np.ascontiguousarray(ev)
for rep in range(3):
    t0 = time.perf_counter()
    h.msm_dev(d.ptr, n) if hasattr(h, "msm_dev") else h.msm(sc)
    print("msm ms", (time.perf_counter() - t0) * 1e3)
