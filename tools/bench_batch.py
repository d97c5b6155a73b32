#!/usr/bin/env python3
"""Dory-shaped workload: k row commitments of n points each over the same bases (src/poly/commitment/dory.zig:646-670), scalars and
results resident in HBM (zg_msm_g1_batch_dev).  usage: bench_batch.py [logn=10] [logk=10]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zolt_amd import api, lib
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 10
logk = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n, k = 1 << logn, 1 << logk
lib.init(0)
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
g = api.generator()
ks = np.zeros((n, 4), dtype=np.uint64); ks[:, 0] = np.arange(1, n + 1, dtype=np.uint64)
xy, _ = lib.g1_scalar_mul_batch(np.repeat(g[None, :], n, axis=0), np.zeros(n, dtype=np.uint8), lib.field_op(lib.FR, lib.OP_TO_MONT, ks))
b = lib.Bases.upload(xy)
raw = np.random.default_rng(2).integers(0, 1 << 63, size=(n * k, 4), dtype=np.uint64)
sc = torch.from_numpy(lib.field_op(lib.FR, lib.OP_TO_MONT, raw).view(np.int64)).to(dev)
out = torch.zeros((k, 9), dtype=torch.int64, device=dev)
for _ in range(2):
    b.msm_batch_dev(sc.data_ptr(), n, k, out.data_ptr(), stream=st.cuda_stream)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    b.msm_batch_dev(sc.data_ptr(), n, k, out.data_ptr(), stream=st.cuda_stream)
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / reps
# spot-check three rows against single MSMs
res = out.cpu().numpy().view(np.uint64)
for j in (0, k // 2, k - 1):
    xy1, inf1 = b.msm_dev(sc.data_ptr() + j * n * 32, n, stream=st.cuda_stream)
    assert int(res[j, 8]) & 0xFF == inf1 and np.array_equal(res[j, :8], xy1)
print(f"{k} x 2^{logn}-point MSMs: {el*1e3:.3f} ms total, {el/k*1e6:.2f} us per MSM, {n*k/el/1e6:.1f} Mpoints/s")
