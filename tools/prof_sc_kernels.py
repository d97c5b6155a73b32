#!/usr/bin/env python3
"""Runs each sumcheck-family kernel at ONE size (default 2^20 entries) a few times, so that rocprofv3's per-kernel averages — durations
with --kernel-trace --stats, counters with --pmc — belong to that size (bench_sumcheck's sessions mix every round's table length under one
kernel name):  sc_sums (session open), sc_fold (one bind of the full table, both layouts), eq_main (eq table), eq_spartan (fused Spartan
open), psc_fold_evals (one product-session round of three tables + its bind).
  rocprofv3 --kernel-trace --stats -- python3 tools/prof_sc_kernels.py 20 8
  rocprofv3 --pmc FETCH_SIZE -- python3 tools/prof_sc_kernels.py 20 8      (one counter group per pass)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zolt_amd import lib  # noqa: E402

v = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = 1 << v
lib.init(0)
rng = np.random.default_rng(7)


def rand_fr(k):
    a = rng.integers(0, 1 << 63, size=(k, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)  # < 2^252 < r: a canonical residue (any value serves: the kernels' work does not depend on it)
    return a


tab = lib.DeviceBuffer.from_host(rand_fr(n))
az, bz, cz = (lib.DeviceBuffer.from_host(rand_fr(n)) for _ in range(3))
out = lib.DeviceBuffer(n * 32)
r = rand_fr(v)
ch = rand_fr(1)[0]
ch128 = ch.copy()
ch128[:2] = 0  # the reference's MontU128Challenge shape: [0, 0, lo, hi]
r128 = r.copy()
r128[:, :2] = 0
for _ in range(reps):
    for layout in (lib.SC_HIGH_HALF, lib.SC_LOW_PAIR):
        s = lib.SumcheckSession.open_dev(tab.ptr, n, layout)  # copy + sc_sums at 2^v
        s.round_sums()
        s.bind(ch if layout == lib.SC_HIGH_HALF else ch128)  # sc_fold at 2^v (full-width / 128-bit challenge)
        s.round_sums()
        s.close()
    lib.fr_eq_table_dev(r, out.ptr)  # eq_main at 2^v (full-width challenges)
    lib.fr_eq_table_dev(r128, out.ptr)  # eq_expand at 2^v (the transcript's 128-bit challenges; from 2^20 entries)
    s = lib.SumcheckSession.open_spartan_dev(r, az.ptr, bz.ptr, cz.ptr)  # eq_spartan at 2^v
    s.round_sums()
    s.close()
lib.sync()
# one product-form session (three tables): evaluations at 2^v, then the fused fold + next evaluations
tabs3 = np.stack([rand_fr(n) for _ in range(3)])
for _ in range(max(1, reps // 4)):
    p = lib.ProductSumcheckSession.open(tabs3)
    p.round_evals((0, 1, 2))
    p.bind(ch128)
    p.round_evals((0, 1, 2))
    p.close()
lib.sync()
print("ok", v, reps)
