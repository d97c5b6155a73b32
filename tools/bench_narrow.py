#!/usr/bin/env python3
"""Full-width against narrow challenges ([0, 0, lo, hi], the reference's stored MontU128Challenge) at the fold sites, host-timed, median
of 5 sessions: usage bench_narrow.py [v=22]. A fold by a narrow challenge takes the 9 x 5-limb product (fp29.hip.h FrMul)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from bench import raw_scalars
    from zolt_amd import lib
    lib.init(0)
    v = int(sys.argv[1]) if len(sys.argv) > 1 else 22
    n = 1 << v
    tab = lib.field_op(lib.FR, lib.OP_TO_MONT, raw_scalars(0x4E415252, 0, n))
    wide = tab[7].copy()
    nar = tab[7].copy()
    nar[:2] = 0
    nar[3] &= np.uint64((1 << 61) - 1)
    d = lib.DeviceBuffer.from_host(tab)
    out = {"v": v}
    # InstructionInputProver's four terms (api.InstructionInputProver): factors 0..7, the two eq tables 8, 9 under (1, g^2) and (g, g^3)
    II_TERMS = [((4, 5, 6, 7), (8, 9), tab[:2], True), ((0, 1, 2, 3), (8, 9), tab[2:4], True)]  # two ZG_PSC_PAIR_SUM terms

    def fold_once(ch):
        s = lib.SumcheckSession.open_dev(d.ptr, n, lib.SC_LOW_PAIR)
        s.round_sums()
        t0 = time.perf_counter()
        s.bind(ch)
        s.round_sums()
        dt = time.perf_counter() - t0
        s.close()
        return dt

    def psc_once(ch, k, spec, points=0xF):
        s = lib.ProductSumcheckSession.open_dev([d.ptr] * k, n)
        s.set_points(points)
        spec(s)
        t0 = time.perf_counter()
        s.bind(ch)
        spec(s)
        dt = time.perf_counter() - t0
        s.close()
        return dt

    def med(f):
        f()
        return round(1e6 * float(np.median([f() for _ in range(5)])), 1)

    for name, ch in (("wide", wide), ("narrow", nar)):
        out[f"session_fold_{name}_us"] = med(lambda: fold_once(ch))
        out[f"psc_fold_evals_p3_{name}_us"] = med(lambda: psc_once(ch, 3, lambda s: s.round_evals((0, 1, 2))))
        out[f"psc_fold_evals_p1q3_{name}_us"] = med(lambda: psc_once(ch, 4, lambda s: s.round_evals((0,), (1, 2, 3), tab[:3])))
        out[f"psc_fold_expr_instruction_input_{name}_us"] = med(lambda: psc_once(ch, 10, lambda s: s.round_expr(
            II_TERMS)))
        out[f"psc_fold_evals_p1q3_points_0_2_{name}_us"] = med(lambda: psc_once(ch, 4, lambda s: s.round_evals((0,), (1, 2, 3), tab[:3]), 0b0101))
        out[f"psc_fold_expr_instruction_input_points_0_2_3_{name}_us"] = med(lambda: psc_once(ch, 10, lambda s: s.round_expr(
            II_TERMS), 0b1101))
        one_first = np.stack([lib.field_op(lib.FR, lib.OP_TO_MONT, np.array([[1, 0, 0, 0]], dtype=np.uint64))[0], tab[1], tab[2]])
        out[f"psc_fold_evals_p1q3_coeff_1_g_g2_points_0_2_{name}_us"] = med(lambda: psc_once(ch, 4, lambda s: s.round_evals((0,), (1, 2, 3), one_first), 0b0101))
        # (a plain fold without a following evaluation is asynchronous: it has no host-visible end to time here — its kernel time is
        # in the rocprofv3 kernel stats, profiles/*_product_form_kernel_stats.csv: psc_fold_kernel)
    print(out)


if __name__ == "__main__":
    main()
