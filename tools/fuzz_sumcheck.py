#!/usr/bin/env python3
"""Randomised differential test of the sumcheck sessions and the device-resident protocol against the CPU oracle (not part of the pytest
suite: run it for as long as you like).  usage: fuzz_sumcheck.py [seconds=60] [seed=1]
Per case: a random table length 2^v (v <= 17) with random or structured entries; a session in either layout, copying or reading the
caller's table in place (zg_sumcheck_open_dev_borrowed), driven through every round with full-width or 128-bit challenges — round sums,
the table after each bind and the final value against the oracle; then runSumcheck resident on the device with a random size of the
LDS tail (ZG_SC_TAIL_MAX) against the oracle's runSumcheck: claim, round polynomials, challenges, final evaluation, result."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import binding as ob  # checker
from zolt_amd import lib

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
lib.init(0)
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def table(n, kind):
    raw = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    t = ob.f_to_mont(ob.FR, raw)
    if kind == 1:  # sparse: mostly zero
        t[rng.random(n) < 0.9] = 0
    elif kind == 2:  # the largest stored values (r - 1, r - 2): lazy sums at their bound
        vals = np.array([[(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)] for v in (R - 1, R - 2)], dtype=np.uint64)
        t[:] = ob.f_to_mont(ob.FR, vals)[rng.integers(0, 2, size=n)]
    elif kind == 3:  # machine words
        t = ob.f_from_u64(ob.FR, rng.integers(0, 1 << 63, size=n, dtype=np.uint64))
    return t


def challenge(narrow):
    c = table(1, 0)[0]
    if narrow:  # the transcript's 128-bit challenges: [0, 0, lo, hi]
        c[:2] = 0
        c[3] &= np.uint64((1 << 61) - 1)
    return c


t0 = time.time()
cases = 0
while time.time() - t0 < budget:
    v = int(rng.choice([0, 1, 2, 3, 5, 6, 7, 9, 11, 12, 13, 14, 16, 17]))
    n = 1 << v
    tab = table(n, int(rng.integers(0, 4)))
    layout = lib.SC_LOW_PAIR if rng.random() < 0.5 else lib.SC_HIGH_HALF
    borrow = rng.random() < 0.5
    d = lib.DeviceBuffer.from_host(tab)
    s = lib.SumcheckSession.open_dev(d.ptr, n, layout, borrow=borrow)
    cur = tab
    for k in range(v):
        want = ob.fr_sum_even_odd(cur) if layout == lib.SC_LOW_PAIR else ob.fr_sum_halves(cur)
        got = s.round_sums()
        assert all(np.array_equal(a, b) for a, b in zip(got, want)), ("sums", v, layout, borrow, k)
        ch = challenge(rng.random() < 0.5)
        s.bind(ch)
        cur = ob.fr_bind_low(cur, ch) if layout == lib.SC_LOW_PAIR else ob.fr_bind_high(cur, ch)
        if rng.random() < 0.3 or len(cur) <= 4:
            assert np.array_equal(s.read().reshape(-1, 4), cur), ("table", v, layout, borrow, k)
    assert np.array_equal(s.final(), cur[0]), ("final", v, layout, borrow)
    s.close()
    assert np.array_equal(d.to_host().reshape(-1, 4), tab), ("caller's table changed", v, layout, borrow)
    os.environ["ZG_SC_TAIL_MAX"] = str(int(rng.choice([1, 2, 64, 128, 1024, 4096])))
    got = lib.run_sumcheck_dev(d.ptr, n)
    claim, rounds, chals, fin, ok = ob.run_sumcheck(tab)
    assert got["result"] == bool(ok) and np.array_equal(got["claim"], claim) and np.array_equal(got["final_eval"], fin), ("run", v, os.environ["ZG_SC_TAIL_MAX"])
    assert np.array_equal(got["rounds"], rounds) and np.array_equal(got["final_point"], chals), ("run rounds", v, os.environ["ZG_SC_TAIL_MAX"])
    d.free()
    cases += 1
print(f"fuzz ok: {cases} tables checked in {time.time() - t0:.1f} s")
