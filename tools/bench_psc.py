#!/usr/bin/env python3
"""Host-timed product-form session kernels at 2^v entries (median of 7): usage bench_psc.py [v=22]; ZG_PSC_BLOCKS sweeps the grid."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from bench import raw_scalars
    from zolt_amd import api, lib
    lib.init(0)
    v = int(sys.argv[1]) if len(sys.argv) > 1 else 22
    n = 1 << v
    tab = lib.field_op(lib.FR, lib.OP_TO_MONT, raw_scalars(0x50534331, 0, n))
    ps = lib.ProductSumcheckSession.open([tab, tab[::-1].copy(), tab, tab])
    g = api.GruenSplitEqPolynomial(tab[400:400 + v])
    d_out, n_out, d_in, n_in = g.getWindowEqTablesDev(1)

    def med(f, reset=True):
        f()
        t = []
        for _ in range(7):
            if reset:  # a repeated call with the same description only collects the mailbox: a Gruen round in between empties it
                ps.round_gruen((0,), d_out, n_out, d_in, n_in)
            t0 = time.perf_counter()
            f()
            t.append(time.perf_counter() - t0)
        return 1e6 * float(np.median(t))

    out = {"v": v, "blocks": os.environ.get("ZG_PSC_BLOCKS", "default"),
           "evals_p2_us": med(lambda: ps.round_evals((0, 1))),
           "evals_p3_us": med(lambda: ps.round_evals((0, 1, 2))),
           "evals_p4_us": med(lambda: ps.round_evals((0, 1, 2, 3))),
           "evals_p1q3_us": med(lambda: ps.round_evals((0,), (1, 2, 3), tab[:3])),
           "evals_p2q2_us": med(lambda: ps.round_evals((0, 1), (2, 3), tab[:2])),
           "gruen_p2_us": med(lambda: ps.round_gruen((0, 1), d_out, n_out, d_in, n_in), reset=False),
           "expr_4x_p2_us": med(lambda: ps.round_expr([((0, 1), (), None), ((2, 3), (), None), ((0, 2), (), None), ((1, 3), (), None)])),
           "expr_instruction_input_shape_us": med(lambda: ps.round_expr([((0, 1), (2, 3), tab[:2]), ((1, 2), (2, 3), tab[:2]), ((0, 3), (2, 3), tab[2:4]),
                                                                         ((1, 3), (2, 3), tab[2:4])]))}
    print(out)
    ps.close()
    g.deinit()


if __name__ == "__main__":
    main()
