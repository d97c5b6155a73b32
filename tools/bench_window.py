#!/usr/bin/env python3
"""MSM latency vs window size c for several n (picks the auto-plan rule). GPU box only."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zolt_amd import api, lib
lib.init(0)
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
nmax = 1 << 19
g = api.generator()
ks = np.zeros((nmax, 4), dtype=np.uint64); ks[:, 0] = np.arange(1, nmax + 1, dtype=np.uint64)
bases_xy, _ = lib.g1_scalar_mul_batch(np.repeat(g[None, :], nmax, axis=0), np.zeros(nmax, dtype=np.uint8), lib.field_op(lib.FR, lib.OP_TO_MONT, ks))
rng = np.random.default_rng(1)
sc_all = torch.from_numpy(lib.field_op(lib.FR, lib.OP_TO_MONT, rng.integers(0, 1 << 63, size=(nmax, 4), dtype=np.uint64)).view(np.int64)).to(dev)
out = torch.zeros(9, dtype=torch.int64, device=dev)
for logn in (10, 12, 14, 16, 17, 18, 19):
    n = 1 << logn
    d_b = torch.from_numpy(bases_xy[:n].view(np.int64)).to(dev)
    row = []
    for c in range(max(4, logn - 6), min(16, logn + 2) + 1):
        b = lib.Bases.upload_dev(d_b.data_ptr(), 0, n, stream=st.cuda_stream, window_bits=c)
        for _ in range(2):
            b.msm_dev_async(sc_all.data_ptr(), n, out.data_ptr(), out[8:].data_ptr(), stream=st.cuda_stream)
        torch.cuda.synchronize()
        reps = 8
        t0 = time.perf_counter()
        for _ in range(reps):
            b.msm_dev_async(sc_all.data_ptr(), n, out.data_ptr(), out[8:].data_ptr(), stream=st.cuda_stream)
        torch.cuda.synchronize()
        row.append((c, (time.perf_counter() - t0) / reps * 1e3))
        b.free()
    best = min(row, key=lambda x: x[1])
    print(f"n=2^{logn}: " + " ".join(f"c{c}={t:.3f}" for c, t in row) + f"  best c={best[0]}")
