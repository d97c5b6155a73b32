#!/usr/bin/env python3
"""Per-round time of the session fold (+ fused next sums) at large table sizes, host-timed: usage bench_fold.py [v=24] [layout=1] [narrow=0]
(narrow=1: challenges of the reference's stored form [0, 0, lo, hi], which take the 9 x 5-limb product).
ZG_SC_MAX_BLOCKS sweeps the grid (default 256 = one workgroup per CU)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from bench import raw_scalars
    from zolt_amd import lib
    lib.init(0)
    v = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    layout = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    narrow = len(sys.argv) > 3 and sys.argv[3] == "1"
    n = 1 << v
    tab = lib.field_op(lib.FR, lib.OP_TO_MONT, raw_scalars(0x464F4C44, 0, min(n, 1 << 20)))
    tab = np.tile(tab, (n // tab.shape[0], 1))
    d = lib.DeviceBuffer.from_host(tab)
    res = {}
    for rep in range(3):
        s = lib.SumcheckSession.open_dev(d.ptr, n, layout)
        s.round_sums()
        for rnd in range(5):
            t0 = time.perf_counter()
            ch = tab[rnd + 1].copy()
            if narrow:
                ch[:2] = 0
                ch[3] &= np.uint64((1 << 61) - 1)
            s.bind(ch)
            s.round_sums()
            dt = time.perf_counter() - t0
            res.setdefault(rnd, []).append(dt)
        s.close()
    out = {"v": v, "layout": "LOW_PAIR" if layout else "HIGH_HALF", "blocks": os.environ.get("ZG_SC_MAX_BLOCKS", "256"),
           "challenge": "narrow [0,0,lo,hi]" if narrow else "full width"}
    for rnd in range(5):
        length = n >> rnd
        us = 1e6 * min(res[rnd])
        out[f"fold_2^{v - rnd}_us"] = round(us, 1)
        out[f"fold_2^{v - rnd}_TBps"] = round(length * 48 / us / 1e6, 2)
    print(out)


if __name__ == "__main__":
    main()
