#!/usr/bin/env python3
"""Latency of small MSMs: (a) own handle per size (auto window), (b) sub-range MSMs on one 2^20-point handle (what
HyperKZG.open's halving commits and HyperKZG.commit of short polynomials do), (c) zg_msm_g1_batch of k short vectors."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zolt_amd import api, lib
lib.init(0)
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
nmax = 1 << 20
g = api.generator()
ks = np.zeros((nmax, 4), dtype=np.uint64); ks[:, 0] = np.arange(1, nmax + 1, dtype=np.uint64)
bases_xy, _ = lib.g1_scalar_mul_batch(np.repeat(g[None, :], nmax, axis=0), np.zeros(nmax, dtype=np.uint8), lib.field_op(lib.FR, lib.OP_TO_MONT, ks))
rng = np.random.default_rng(1)
raw = rng.integers(0, 1 << 63, size=(nmax, 4), dtype=np.uint64)
sc_host = lib.field_op(lib.FR, lib.OP_TO_MONT, raw)
sc_all = torch.from_numpy(sc_host.view(np.int64)).to(dev)
out = torch.zeros(9, dtype=torch.int64, device=dev)

def timeit(b, n, reps=20):
    for _ in range(3):
        b.msm_dev_async(sc_all.data_ptr(), n, out.data_ptr(), out[8:].data_ptr(), stream=st.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        b.msm_dev_async(sc_all.data_ptr(), n, out.data_ptr(), out[8:].data_ptr(), stream=st.cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

d_b = torch.from_numpy(bases_xy.view(np.int64)).to(dev)
for logn in (4, 6, 8, 10, 12, 14, 16):
    n = 1 << logn
    b = lib.Bases.upload_dev(d_b.data_ptr(), 0, n, stream=st.cuda_stream)
    print(f"(a) own handle   n=2^{logn:2d}: {timeit(b, n):8.3f} ms per MSM (serial, one stream)")
    b.free()
big = lib.Bases.upload_dev(d_b.data_ptr(), 0, nmax, stream=st.cuda_stream)
for logn in (4, 8, 10, 12, 14, 16, 18, 19, 20):
    print(f"(b) 2^20 handle  n=2^{logn:2d}: {timeit(big, 1 << logn):8.3f} ms per MSM (serial, one stream)")
big.free()
for logn, k in ((10, 64), (12, 64), (14, 16)):
    n = 1 << logn
    b = lib.Bases.upload(bases_xy[:n])
    batches = [sc_host[i * n:(i + 1) * n] for i in range(k)]
    b.msm_batch(batches)
    t0 = time.perf_counter()
    b.msm_batch(batches)
    el = time.perf_counter() - t0
    print(f"(c) batch of {k:3d} x 2^{logn}: {el*1e3:8.3f} ms total, {el/k*1e3:7.3f} ms per MSM (host scalars, incl. H2D)")
    b.free()
