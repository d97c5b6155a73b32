import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from zolt_amd import api, lib
n = 20; T = 1 << n
lib.init(0)
rng = np.random.default_rng(n)
wm = np.zeros((T, 43, 4), dtype=np.uint64)
wm[:, :, 0] = rng.integers(0, 1 << 62, size=(T, 43), dtype=np.uint64)
mont = lambda k: lib.field_op(lib.FR, lib.OP_TO_MONT, rng.integers(0, 1 << 62, size=(k, 4), dtype=np.uint64))
ro, rp, ch, g = mont(n), mont(n), mont(n), mont(8)
d_rows = lib.DeviceBuffer.from_host(wm)
for rep in range(2):
    t0 = time.perf_counter()
    p = api.Stage3Prover(None, ro, rp, g[:5], g[5], g[6], g[:3], g[3:6], d_rows=d_rows.ptr)
    t1 = time.perf_counter()
    ts = []
    for k in range(n):
        a = time.perf_counter(); p.computeRoundPolynomial(); b = time.perf_counter(); p.bindChallenge(ch[k]); c = time.perf_counter()
        ts.append((round((b - a) * 1e3, 3), round((c - b) * 1e3, 3)))
    p.deinit()
print("build ms", round((t1 - t0) * 1e3, 2)); print(ts)
