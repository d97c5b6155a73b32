#!/usr/bin/env python3
"""HyperKZG.setup's fixed-base batch (src/poly/commitment/mod.zig:194-199): tau^i * G for i < n with the fixed-base kernel
against the generic double-and-add kernel; the reference's captured run took 35 578 ms for n = 1280 (logs/zolt.log:10-12).
Prints one JSON line. Host buffers in and out (PCIe included)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from zolt_amd import api, lib
    lib.init(0)
    g = api.generator()
    out = {}
    for n in (1280, 1 << 16, 1 << 20, 1 << 24):
        raw = np.zeros((n, 4), dtype=np.uint64)
        raw[:, 0] = np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(1)
        raw[:, 1] = np.arange(n, dtype=np.uint64) * np.uint64(0xBF58476D1CE4E5B9)
        raw[:, 2] = np.arange(n, dtype=np.uint64) * np.uint64(0x94D049BB133111EB)
        raw[:, 3] = np.arange(n, dtype=np.uint64) & np.uint64(0x0FFFFFFFFFFFFFFF)
        sc = lib.field_op(lib.FR, lib.OP_TO_MONT, raw)
        lib.g1_fixed_base_mul_batch(g, sc[:1024])
        t0 = time.perf_counter()
        fx, fi = lib.g1_fixed_base_mul_batch(g, sc)
        t_fixed = time.perf_counter() - t0
        rec = {"fixed_base_ms": t_fixed * 1e3, "points_per_s": n / t_fixed}
        if n <= 1 << 20:
            bx = np.repeat(g[None, :], n, axis=0)
            bi = np.zeros(n, dtype=np.uint8)
            lib.g1_scalar_mul_batch(bx[:1024], bi[:1024], sc[:1024])
            t0 = time.perf_counter()
            gx, gi = lib.g1_scalar_mul_batch(bx, bi, sc)
            t_gen = time.perf_counter() - t0
            assert np.array_equal(fx, gx) and np.array_equal(fi, gi)
            rec["double_and_add_ms"] = t_gen * 1e3
            rec["speedup"] = t_gen / t_fixed
        out[str(n)] = rec
    out["reference_cpu_setup_1280_ms"] = 35578.43
    print(json.dumps(out))


if __name__ == "__main__":
    main()
