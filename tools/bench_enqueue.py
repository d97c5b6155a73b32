#!/usr/bin/env python3
"""Host enqueue time vs GPU completion time of pipelined MSMs (is a small MSM launch-bound on the host?). GPU box only.
usage: bench_enqueue.py [logn ...]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zolt_amd import api, lib
lib.init(0)
dev = torch.device("cuda", 0)
streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
torch.cuda.set_stream(streams[0])
g = api.generator()
for logn in [int(a) for a in sys.argv[1:]] or [14, 17, 20]:
    n = 1 << logn
    ks = np.zeros((n, 4), dtype=np.uint64); ks[:, 0] = np.arange(1, n + 1, dtype=np.uint64)
    bxy, _ = lib.g1_scalar_mul_batch(np.repeat(g[None, :], n, axis=0), np.zeros(n, dtype=np.uint8), lib.field_op(lib.FR, lib.OP_TO_MONT, ks))
    d_b = torch.from_numpy(bxy.view(np.int64)).to(dev)
    rng = np.random.default_rng(1)
    sc = torch.from_numpy(lib.field_op(lib.FR, lib.OP_TO_MONT, rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)).view(np.int64)).to(dev)
    b = lib.Bases.upload_dev(d_b.data_ptr(), 0, n, stream=streams[0].cuda_stream)
    out = torch.zeros((64, 9), dtype=torch.int64, device=dev)
    for i in range(6):
        b.msm_dev_async(sc.data_ptr(), n, out[i].data_ptr(), out[i, 8:].data_ptr(), stream=streams[i % 3].cuda_stream)
    torch.cuda.synchronize()
    reps = 48
    t0 = time.perf_counter()
    for i in range(reps):
        b.msm_dev_async(sc.data_ptr(), n, out[i].data_ptr(), out[i, 8:].data_ptr(), stream=streams[i % 3].cuda_stream)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"n=2^{logn}: host enqueue {1e3 * (t1 - t0) / reps:.3f} ms/MSM, complete {1e3 * (t2 - t0) / reps:.3f} ms/MSM")
    b.free()
