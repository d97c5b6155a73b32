#!/usr/bin/env python3
"""zg_msm_g1 with HOST scalars (the entry point an unmodified MSM.compute / HyperKZG.commit call site reaches), sliced vs
unsliced (ZG_MSM_HOST_SLICES), against the device-resident call. Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from bench import SEED, closed_form_scalar, raw_scalars
    from zolt_amd import api, lib
    lib.init(0)
    logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    n = 1 << logn
    g = api.generator()
    ks = np.zeros((n, 4), dtype=np.uint64)
    ks[:, 0] = np.arange(1, n + 1, dtype=np.uint64)
    xy, _ = lib.g1_scalar_mul_batch(np.repeat(g[None, :], n, axis=0), np.zeros(n, dtype=np.uint8), lib.field_op(lib.FR, lib.OP_TO_MONT, ks))
    raw = raw_scalars(SEED, 0, n)
    sc = lib.field_op(lib.FR, lib.OP_TO_MONT, raw)
    want = api.MSM.scalarMul(g, api.fr_from_int(closed_form_scalar(raw, 0)))
    out = {"points": n}
    b = lib.Bases.upload(xy)
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    for _ in range(3):
        r = b.msm_dev(d_sc.data_ptr(), n)
    assert r[1] == want[1] and np.array_equal(r[0], want[0])
    t0 = time.perf_counter()
    for _ in range(10):
        b.msm_dev(d_sc.data_ptr(), n)
    out["device_scalars_ms"] = (time.perf_counter() - t0) / 10 * 1e3
    for slices in (1, 2, 4, 8):
        os.environ["ZG_MSM_HOST_SLICES"] = str(slices)
        for _ in range(2):
            r = b.msm(sc)
        assert r[1] == want[1] and np.array_equal(r[0], want[0]), slices
        t0 = time.perf_counter()
        for _ in range(10):
            b.msm(sc)
        out[f"host_scalars_slices{slices}_ms"] = (time.perf_counter() - t0) / 10 * 1e3
    print(json.dumps(out))


if __name__ == "__main__":
    main()
