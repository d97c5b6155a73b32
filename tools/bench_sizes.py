#!/usr/bin/env python3
"""MSM latency across sizes 2^10..2^22 (uniform scalars, single stream) — the sizes HyperKZG.open walks through."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zolt_amd import api, lib
lib.init(0)
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
nmax = 1 << 22
g = api.generator()
ks = np.zeros((nmax, 4), dtype=np.uint64); ks[:, 0] = np.arange(1, nmax + 1, dtype=np.uint64)
bases_xy, _ = lib.g1_scalar_mul_batch(np.repeat(g[None, :], nmax, axis=0), np.zeros(nmax, dtype=np.uint8), lib.field_op(lib.FR, lib.OP_TO_MONT, ks))
rng = np.random.default_rng(1)
raw = rng.integers(0, 1 << 63, size=(nmax, 4), dtype=np.uint64)
sc_all = torch.from_numpy(lib.field_op(lib.FR, lib.OP_TO_MONT, raw).view(np.int64)).to(dev)
out = torch.zeros(9, dtype=torch.int64, device=dev)
for logn in (10, 12, 14, 16, 18, 20, 22):
    n = 1 << logn
    d_b = torch.from_numpy(bases_xy[:n].view(np.int64)).to(dev)
    b = lib.Bases.upload_dev(d_b.data_ptr(), 0, n, stream=st.cuda_stream)
    for _ in range(2):
        b.msm_dev_async(sc_all.data_ptr(), n, out.data_ptr(), out[8:].data_ptr(), stream=st.cuda_stream)
    torch.cuda.synchronize()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        b.msm_dev_async(sc_all.data_ptr(), n, out.data_ptr(), out[8:].data_ptr(), stream=st.cuda_stream)
    torch.cuda.synchronize()
    print(f"n=2^{logn:2d}: {(time.perf_counter()-t0)/reps*1e3:8.3f} ms per MSM")
    b.free()
