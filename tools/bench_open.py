#!/usr/bin/env python3
"""HyperKZG.open / batchOpen wall time (host buffers in, proof out) at a few sizes."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zolt_amd import api, lib
lib.init(0)
rng = np.random.default_rng(3)
for v in (12, 16, 20):
    n = 1 << v
    g = api.generator()
    ks = np.zeros((n, 4), dtype=np.uint64); ks[:, 0] = np.arange(1, n + 1, dtype=np.uint64)
    xy, inf = lib.g1_scalar_mul_batch(np.repeat(g[None, :], n, axis=0), np.zeros(n, dtype=np.uint8), lib.field_op(lib.FR, lib.OP_TO_MONT, ks))
    params = api.HyperKZG.SetupParams(xy, inf)
    ev = lib.field_op(lib.FR, lib.OP_TO_MONT, rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64))
    pt = lib.field_op(lib.FR, lib.OP_TO_MONT, rng.integers(0, 1 << 63, size=(v, 4), dtype=np.uint64))
    api.HyperKZG.open(params, ev, pt, np.zeros(4, dtype=np.uint64))
    t0 = time.perf_counter()
    for _ in range(3):
        q, fin = api.HyperKZG.open(params, ev, pt, np.zeros(4, dtype=np.uint64))
    t_open = (time.perf_counter() - t0) / 3 * 1e3
    t0 = time.perf_counter()
    c = api.HyperKZG.commit(params, ev)
    t_commit = (time.perf_counter() - t0) * 1e3
    print(f"v={v}: open {t_open:8.3f} ms   (one commit of the same table with host scalars: {t_commit:.3f} ms)")
    params.deinit()
