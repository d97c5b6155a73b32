#!/usr/bin/env python3
"""Single short MSMs (the sizes `zolt prove` commits at its default trace length): latency per call for n points on a handle of
`handle_n` points; run under rocprofv3 --kernel-trace for the per-kernel timeline. usage: bench_small_msm.py [n=1024] [handle_n=n]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from bench import raw_scalars
    from oracle import binding as ob  # bases only (generator multiples)
    from zolt_amd import lib
    lib.init(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    hn = int(sys.argv[2]) if len(sys.argv) > 2 else n
    gm = ob.g1_gen_multiples(hn)
    sc = lib.field_op(lib.FR, lib.OP_TO_MONT, raw_scalars(0x534D414C, 0, n))
    h = lib.Bases.upload(gm) if hasattr(lib, "Bases") else None
    d = lib.DeviceBuffer.from_host(sc)
    plan = h.plan() if hasattr(h, "plan") else None
    for _ in range(5):
        h.msm_dev(d.ptr, n)
    t = []
    for _ in range(30):
        t0 = time.perf_counter()
        h.msm_dev(d.ptr, n)
        t.append(time.perf_counter() - t0)
    print({"n": n, "handle_n": hn, "plan": plan, "us_per_msm_median": 1e6 * float(np.median(t)), "us_min": 1e6 * min(t)})


if __name__ == "__main__":
    main()
