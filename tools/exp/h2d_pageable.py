#!/usr/bin/env python3
"""What a pageable host-to-device copy costs and what it leaves behind: zg_memcpy_h2d of `mb` MB from (a) one long-lived array, (b) a fresh
array per call, (c) memory from zg_host_alloc (pinned); after each copy a small unrelated call (a 64-byte H2D + a 1 MB pool allocation)
is timed too — round 5 saw 15 ms of deferred cost land in the call AFTER a copy from long-lived pageable memory."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from zolt_amd import lib
lib.init(0)
L = lib._lib
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = mb << 20
d = lib.DeviceBuffer(n)
small = lib.DeviceBuffer(64)
tiny = np.zeros(8, dtype=np.uint64)


def after():
    t0 = time.perf_counter()
    L.zg_memcpy_h2d(C.c_void_p(small.ptr), tiny.ctypes.data_as(C.c_void_p), C.c_size_t(64))
    b = lib.DeviceBuffer(1 << 20)
    b.free()
    big = lib.DeviceBuffer(1 << 30)  # a pool miss the first time: a real hipMalloc, as a proving key's table is
    big.free()
    return (time.perf_counter() - t0) * 1e3


def run(name, get):
    ts, af = [], []
    for rep in range(6):
        a = get()
        t0 = time.perf_counter()
        L.zg_memcpy_h2d(C.c_void_p(d.ptr), C.c_void_p(a if isinstance(a, int) else a.ctypes.data), C.c_size_t(n))
        ts.append((time.perf_counter() - t0) * 1e3)
        af.append(after())
    print(f"{name:34s} copy ms {[round(x, 2) for x in ts]}  next calls ms {[round(x, 2) for x in af]}", flush=True)


long_lived = np.ones(n // 8, dtype=np.uint64)
run("long-lived pageable array", lambda: long_lived)
run("fresh pageable array per call", lambda: np.ones(n // 8, dtype=np.uint64))
p = C.c_void_p()
assert L.zg_host_alloc(C.c_size_t(n), C.byref(p)) == 0
C.memset(p, 1, n)
run("pinned (zg_host_alloc)", lambda: p.value)
run("long-lived pageable array (again)", lambda: long_lived)
