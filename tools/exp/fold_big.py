#!/usr/bin/env python3
"""One fold of a 2^v-entry session table (32 B read per entry, 16 B written, next sums fused), host-timed bind + round_sums, best of 5:
python3 tools/exp/fold_big.py [v ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from zolt_amd import lib
lib.init(0)
rng = np.random.default_rng(1)
for v in [int(a) for a in sys.argv[1:]] or [24, 25, 26]:
    n = 1 << v
    d = lib.DeviceBuffer(n * 32)
    lib._lib.zg_dev_memset(__import__("ctypes").c_void_p(d.ptr), 1, __import__("ctypes").c_size_t(n * 32))
    ch = rng.integers(0, 1 << 62, size=4, dtype=np.uint64); ch[3] >>= 4
    for layout, name in ((lib.SC_HIGH_HALF, "HIGH_HALF"), (lib.SC_LOW_PAIR, "LOW_PAIR")):
        best = None
        for rep in range(5):
            s = lib.SumcheckSession.open_dev(d.ptr, n, layout, borrow=True)
            s.round_sums()
            t0 = time.perf_counter()
            s.bind(ch)
            s.round_sums()
            dt = time.perf_counter() - t0
            s.close()
            best = dt if best is None else min(best, dt)
        print(f"2^{v} {name}: {best * 1e6:.1f} us, {n * 48 / best / 1e12:.2f} TB/s", flush=True)
    d.free()
