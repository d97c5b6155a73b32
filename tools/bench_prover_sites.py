#!/usr/bin/env python3
"""Launches the round-2 kernels at sizes where their bound shows (run under `rocprofv3 --kernel-trace --stats` for the per-kernel
durations in DESIGN.md's table): RAF cubic round sums + LowToHigh fold at 2^22 entries, Lasso bit-split sums over 2^22 lookups,
fixed-base batch at 2^20 scalars, affine add at 2^20 pairs, eq table + Spartan combine at 2^22."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from bench import raw_scalars
    from zolt_amd import api, lib
    lib.init(0)
    v = 22
    n = 1 << v
    tab = lib.field_op(lib.FR, lib.OP_TO_MONT, raw_scalars(0x52414631, 0, n))
    for _ in range(3):
        s = lib.SumcheckSession.open(tab, lib.SC_LOW_PAIR)
        s.raf_round(api.fr_from_int(0x7FFF8000), 8)
        s.bind(tab[5])
        s.raf_round(api.fr_from_int(0x7FFF8000 + 8 * 12345), 16)
        s.close()
    idx = raw_scalars(0x4C415353, 0, n // 2).reshape(n, 2)
    d_eq, d_idx = lib.DeviceBuffer.from_host(tab), lib.DeviceBuffer.from_host(idx)
    for bit in (0, 17, 70):
        lib.fr_bit_split_sums_dev(d_eq.ptr, d_idx.ptr, n, bit)
    d_eq.free(); d_idx.free()
    m = 1 << 20
    g = api.generator()
    sc = tab[:m]
    for _ in range(2):
        xy, inf = lib.g1_fixed_base_mul_batch(g, sc)
    for _ in range(2):
        lib.g1_affine_add_batch(xy[:m // 2], None, xy[m // 2:], None)
    r = tab[:v]
    d_out = lib.DeviceBuffer(n * 32)
    d_a = lib.DeviceBuffer.from_host(tab)
    for _ in range(3):
        lib.fr_eq_table_dev(r, d_out.ptr)
        lib.sync()
        lib.fr_spartan_combine_dev(d_out.ptr, d_a.ptr, d_a.ptr, d_a.ptr, n, d_out.ptr)
        lib.sync()
    # LassoProver: 2^22 cycles, 16 address rounds (scale by index bit + next round's sums in one pass), then 22 cycle rounds
    import time
    w = tab[100:100 + v]
    idx16 = idx.copy()
    idx16[:, 0] &= np.uint64(0xFFFF)
    idx16[:, 1] = 0
    lp = api.LassoProver(idx16, v, 16, w)
    lib.sync()
    t = []
    for rnd in range(16 + v):
        t0 = time.perf_counter()
        lp.computeRoundPolynomial()
        lp.receiveChallenge(tab[200 + rnd])
        t.append(time.perf_counter() - t0)
    lp.deinit()
    print(f"lasso 2^{v} cycles: address round {1e6 * np.median(t[1:16]):.1f} us (first, unfused sums: {1e6 * t[0]:.1f}), "
          f"cycle rounds {1e6 * np.sum(t[16:]):.0f} us for all {v}, whole protocol {1e3 * np.sum(t):.2f} ms")
    # the size the reference's own run has (log_T = 13, logs/zolt.log): latency per round, not bandwidth
    lp = api.LassoProver(idx16[:1 << 13], 13, 16, tab[100:113])
    t = []
    for rnd in range(16 + 13):
        t0 = time.perf_counter()
        lp.computeRoundPolynomial()
        lp.receiveChallenge(tab[200 + rnd])
        t.append(time.perf_counter() - t0)
    lp.deinit()
    print(f"lasso 2^13 cycles: address round {1e6 * np.median(t[1:16]):.1f} us, cycle round {1e6 * np.median(t[16:]):.1f} us")
    # product-form sessions at 2^22 entries: three-table cubic evaluations, two-table Gruen sums, the fold of all tables
    ps = lib.ProductSumcheckSession.open([tab, tab[::-1].copy(), tab])
    ps.round_evals((0, 1, 2))
    ps.round_evals((0,), (1, 2), tab[:2])
    g = api.GruenSplitEqPolynomial(tab[400:400 + v])
    d_out, n_out, d_in, n_in = g.getWindowEqTablesDev(1)
    ps.round_gruen((0, 1), d_out, n_out, d_in, n_in)
    ps.bind(tab[9])
    ps.round_evals((0, 1, 2))
    ps.close()
    g.deinit()
    # R1CSInputEvaluator.computeClaimedInputs: 36 column MLEs from a 2^20-cycle witness matrix (1.2 GB, resident)
    T = 1 << 20
    rows = np.tile(tab[:T // 4], (36 * 4, 1))[:T * 36].reshape(T, 36, 4)
    d_rows = lib.DeviceBuffer.from_host(rows)
    lib.fr_rows_mle_dev(d_rows.ptr, T, 36, tab[500:520])
    t0 = time.perf_counter()
    for _ in range(5):
        lib.fr_rows_mle_dev(d_rows.ptr, T, 36, tab[500:520])
    dt = (time.perf_counter() - t0) / 5
    print(f"computeClaimedInputs 2^20 cycles x 36 inputs: {1e6 * dt:.0f} us ({T * 36 * 32 / dt / 1e9:.0f} GB/s of matrix)")
    d_rows.free()
    # GruenSplitEqPolynomial init: both halves' prefix-table sets for a 24-variable tau (m = 12)
    t0 = time.perf_counter()
    for _ in range(20):
        api.GruenSplitEqPolynomial(tab[300:324])
    print(f"GruenSplitEqPolynomial.init, 24 variables: {1e6 * (time.perf_counter() - t0) / 20:.1f} us")
    print("ok")


if __name__ == "__main__":
    main()
