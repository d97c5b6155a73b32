#!/usr/bin/env python3
"""Stage 3's prefix / suffix provers (api.ShiftPrefixSuffixProver, api.RegistersPrefixSuffixProver) at a given trace length, the witness
matrix resident in HBM (as Stage 1 leaves it): time to build (affine maps of the witness rows, weighted column sums, prefix / suffix
tables) and time for all rounds with a stand-in challenge per round.   python tools/bench_stage3.py [log_t=20]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zolt_amd import api, lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
T = 1 << n
lib.init(0)
rng = np.random.default_rng(n)
wm = np.zeros((T, 43, 4), dtype=np.uint64)
wm[:, :, 0] = rng.integers(0, 1 << 62, size=(T, 43), dtype=np.uint64)
mont = lambda k: lib.field_op(lib.FR, lib.OP_TO_MONT, rng.integers(0, 1 << 62, size=(k, 4), dtype=np.uint64))
ro, rp, ch, g = mont(n), mont(n), mont(n), mont(6)
t0 = time.perf_counter(); d_rows = lib.DeviceBuffer.from_host(wm); res = {"log_t": n, "witness_upload_ms": round((time.perf_counter() - t0) * 1e3, 2)}
for name, make in (("shift", lambda: api.ShiftPrefixSuffixProver(None, ro, rp, g[:5], d_rows=d_rows.ptr)), ("registers", lambda: api.RegistersPrefixSuffixProver(None, ro, g[5], d_rows=d_rows.ptr))):
    for rep in range(2):
        t0 = time.perf_counter(); p = make(); t1 = time.perf_counter()
        claim = g[0]
        for k in range(n):
            p.computeRoundEvals(claim)
            p.bind(ch[k])
        t2 = time.perf_counter()
        p.deinit()
    res[name] = {"init_ms": round((t1 - t0) * 1e3, 2), "rounds_ms": round((t2 - t1) * 1e3, 2)}
print(json.dumps(res))
