#!/usr/bin/env python3
"""HyperKZG.batchCommit of k full-size polynomials through the host entry point (zg_msm_g1_batch): interleaved copies against k
separate zg_msm_g1 calls. Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from bench import SEED, raw_scalars
    from zolt_amd import api, lib
    lib.init(0)
    n, k = 1 << 20, 3
    g = api.generator()
    ks = np.zeros((n, 4), dtype=np.uint64)
    ks[:, 0] = np.arange(1, n + 1, dtype=np.uint64)
    xy, _ = lib.g1_scalar_mul_batch(np.repeat(g[None, :], n, axis=0), np.zeros(n, dtype=np.uint8), lib.field_op(lib.FR, lib.OP_TO_MONT, ks))
    vecs = [lib.field_op(lib.FR, lib.OP_TO_MONT, raw_scalars(SEED + j, 0, n)) for j in range(k)]
    b = lib.Bases.upload(xy)
    singles = [b.msm(v) for v in vecs]
    out, inf = b.msm_batch(vecs)
    for j in range(k):
        assert inf[j] == singles[j][1] and np.array_equal(out[j], singles[j][0])
    t0 = time.perf_counter()
    for _ in range(5):
        b.msm_batch(vecs)
    t_batch = (time.perf_counter() - t0) / 5 * 1e3
    t0 = time.perf_counter()
    for _ in range(5):
        for v in vecs:
            b.msm(v)
    t_sep = (time.perf_counter() - t0) / 5 * 1e3
    print(json.dumps({"points": n, "vectors": k, "batch_ms": t_batch, "separate_calls_ms": t_sep}))


if __name__ == "__main__":
    main()
