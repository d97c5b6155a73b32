#!/usr/bin/env python3
"""What slicing one 2^20-point MSM costs by itself (scalars already resident: no copy to hide): k slices as partial MSMs over
bases[off..] on three rotating streams + one combine, against the unsliced call. Separates the slicing overhead of zg_msm_g1's
host path from the H2D copy it is meant to hide."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from bench import SEED, raw_scalars
    from zolt_amd import api, lib
    lib.init(0)
    n = 1 << 20
    g = api.generator()
    ks = np.zeros((n, 4), dtype=np.uint64)
    ks[:, 0] = np.arange(1, n + 1, dtype=np.uint64)
    xy, _ = lib.g1_scalar_mul_batch(np.repeat(g[None, :], n, axis=0), np.zeros(n, dtype=np.uint8), lib.field_op(lib.FR, lib.OP_TO_MONT, ks))
    sc = lib.field_op(lib.FR, lib.OP_TO_MONT, raw_scalars(SEED, 0, n))
    b = lib.Bases.upload(xy)
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    streams = [torch.cuda.Stream() for _ in range(3)]
    ref = b.msm_dev(d_sc.data_ptr(), n, stream=streams[0].cuda_stream)
    out = {}
    for k in (1, 2, 4, 8):
        d_parts = torch.zeros((k, 12), dtype=torch.int64, device="cuda")
        per = (n + k - 1) // k

        def run():
            for i in range(k):
                a = i * per
                cnt = min(per, n - a)
                b.msm_partial_fast_dev(d_sc.data_ptr() + 32 * a, cnt, d_parts[i].data_ptr(), off=a, stream=streams[i % 3].cuda_stream)
            torch.cuda.synchronize()
            return lib.combine_partials_dev(d_parts.data_ptr(), k, stream=streams[0].cuda_stream)
        for _ in range(3):
            got = run()
        assert got[1] == ref[1] and np.array_equal(got[0], ref[0])
        t0 = time.perf_counter()
        for _ in range(10):
            run()
        out[f"slices{k}_ms"] = (time.perf_counter() - t0) / 10 * 1e3
    print(json.dumps(out))


if __name__ == "__main__":
    main()
