#!/usr/bin/env python3
"""Times the 2^20-point MSM on non-uniform scalar distributions (0/1 flags, bytes, constants) — the shapes
real witness columns have — to show the accumulate scheduling is skew-robust. GPU box only."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zolt_amd import api, lib  # noqa: E402

lib.init(0)
dev = torch.device("cuda", 0)
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << logn
g = api.generator()
ks = np.zeros((n, 4), dtype=np.uint64)
ks[:, 0] = np.arange(1, n + 1, dtype=np.uint64)
bases_xy, _ = lib.g1_scalar_mul_batch(np.repeat(g[None, :], n, axis=0), np.zeros(n, dtype=np.uint8), lib.field_op(lib.FR, lib.OP_TO_MONT, ks))
d_bases = torch.from_numpy(bases_xy.view(np.int64)).to(dev)
st = torch.cuda.current_stream().cuda_stream
bases = lib.Bases.upload_dev(d_bases.data_ptr(), 0, n, stream=st)
rng = np.random.default_rng(3)
cases = {
    "uniform_254bit": rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64),
    "boolean": np.stack([rng.integers(0, 2, size=n, dtype=np.uint64)] + [np.zeros(n, dtype=np.uint64)] * 3, axis=1),
    "bytes": np.stack([rng.integers(0, 256, size=n, dtype=np.uint64)] + [np.zeros(n, dtype=np.uint64)] * 3, axis=1),
    "u32": np.stack([rng.integers(0, 1 << 32, size=n, dtype=np.uint64)] + [np.zeros(n, dtype=np.uint64)] * 3, axis=1),
    "all_equal": np.tile(np.array([[0x123456789ABCDEF, 77, 0, 0]], dtype=np.uint64), (n, 1)),
    "all_ones": np.tile(np.array([[1, 0, 0, 0]], dtype=np.uint64), (n, 1)),
}
out = torch.zeros(9, dtype=torch.int64, device=dev)
for name, raw in cases.items():
    sc = torch.from_numpy(lib.field_op(lib.FR, lib.OP_TO_MONT, np.ascontiguousarray(raw)).view(np.int64)).to(dev)
    for _ in range(2):
        bases.msm_dev_async(sc.data_ptr(), n, out.data_ptr(), out[8:].data_ptr(), stream=st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        bases.msm_dev_async(sc.data_ptr(), n, out.data_ptr(), out[8:].data_ptr(), stream=st)
    torch.cuda.synchronize()
    print(f"{name:16s} {(time.perf_counter() - t0) / reps * 1e3:9.3f} ms per 2^{logn} MSM (sched={os.environ.get('ZG_MSM_CHUNK_SCHED', '1')})")
