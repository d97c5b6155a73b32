#!/usr/bin/env python3
"""Time of zg_g1_bases_upload_dev (plan + workspaces + table of multiples) at 2^v resident bases, warm pool: python3 tools/exp/table_build.py [v ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from zolt_amd import lib, api
lib.init(0)
for v in [int(a) for a in sys.argv[1:]] or [20, 22]:
    n = 1 << v
    h, _, _ = lib.Bases.hyperkzg_setup(api.generator(), api.fr_from_int(0x12345678), n, want_points=False, expected_uses=1)  # the points, no table
    del h
    for v1 in ("0", "1"):
        os.environ["ZG_MSM_PRECOMPUTE_V1"] = v1
        ts = []
        for rep in range(3):
            t0 = time.perf_counter()
            h, _, _ = lib.Bases.hyperkzg_setup(api.generator(), api.fr_from_int(0x12345678), n, want_points=False)
            ts.append((time.perf_counter() - t0) * 1e3)
            tb = h.table_bytes()
            h.free()
        print(f"2^{v} powers, ZG_MSM_PRECOMPUTE_V1={v1}: HyperKZG.setup with the table {min(ts):.2f} ms (table {tb / 1e9:.2f} GB)", flush=True)
