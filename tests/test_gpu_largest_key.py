"""Parity at the reference's largest proving key: srs_size = 256 + nextPowerOfTwo(trace_len), up to 2^24 + 256 points
(src/host/mod.zig:384-387; the key is HyperKZG.setup's tau^i * G, src/poly/commitment/mod.zig:174-213). Until round 6 the largest MSM
a -m gpu test held against a closed form was 2^23 points; 2^24 appeared only in timing scripts. Here, through the C ABI:

  * MSM over 2^24 bases (i+1) G with uniform scalars, on the 16 GiB table of precomputed multiples AND on a table-less handle
    (zg_msm_config.expected_uses = 1, the mode one `zolt prove` run should ask for) == the closed form (sum s_i (i+1) mod r) G;
  * zg_hyperkzg_setup(2^24 + 256): sampled powers (both sides of the 2^8 / 2^16 / 2^24 table boundaries, random indices, the LAST
    power) == scalarMul(G, tau^i mod r), with tau^i from Python's pow;
  * one commitment from machine words (zg_msm_g1_u64, what zkvm/mod.zig:1538-1607 commits) and one from full-width scalars over that
    key, on both kinds of handle == (sum w_i tau^i mod r) G, the sum by Horner's rule in Python integers.

Size-independent properties only (the C oracle needs ~3 minutes per 2^24-point MSM); sizes the oracle can finish are
tests/test_gpu_msm.py. The timings the test prints (table bytes, build time) go to the bench's side file as well."""
import time

import numpy as np
import pytest

from tests import util as U

pytestmark = pytest.mark.gpu
N24 = 1 << 24
N_KEY = N24 + 256  # src/host/mod.zig:384-387 at trace_len = 2^24


@pytest.fixture(scope="module")
def zl():
    from zolt_amd import lib
    lib.init()
    return lib


def _weighted_sum(raw, first):
    """sum_i raw_i * (first + i) over (n, 4) u64 limbs, exact: 32-bit halves times the (< 2^25) weights summed in u64 over rows of 64
    (57 + 6 bits), the row sums added as Python integers"""
    n = raw.shape[0]
    assert n % 64 == 0 and first + n < 1 << 25
    w = np.arange(first, first + n, dtype=np.uint64)
    tot = 0
    for limb in range(4):
        for half, shift in ((raw[:, limb] & np.uint64(0xFFFFFFFF), 0), (raw[:, limb] >> np.uint64(32), 32)):
            rows = (half * w).reshape(-1, 64).sum(axis=1, dtype=np.uint64)
            tot += sum(rows.tolist()) << (64 * limb + shift)
    return tot


def _horner(values, tau, mod):
    """sum_i values[i] * tau^i mod `mod` (values: Python ints)"""
    acc = 0
    for v in reversed(values):
        acc = (acc * tau + v) % mod
    return acc


def _ints256(raw):
    a, b, c, d = (raw[:, k].tolist() for k in range(4))
    return [w | (x << 64) | (y << 128) | (z << 192) for w, x, y, z in zip(a, b, c, d)]


def test_msm_2e24_closed_form_with_the_table_and_without(zl):
    from oracle import pymodel as pm
    from zolt_amd import api
    n, g = N24, api.generator()
    ks = np.zeros((n, 4), dtype=np.uint64)
    ks[:, 0] = np.arange(1, n + 1, dtype=np.uint64)
    bases, binf = zl.g1_fixed_base_mul_batch(g, zl.field_op(zl.FR, zl.OP_TO_MONT, ks))  # (i+1) G, src/bench.zig:261-268
    del ks
    assert not binf.any()
    raw = U.random_raw256(0x5A4F4C54, n)
    sc = zl.field_op(zl.FR, zl.OP_TO_MONT, raw)
    want = pm.ec_mul(_weighted_sum(raw, 1) % pm.R_MOD, pm.G1)
    del raw
    results = {}
    for uses in (0, 1):
        t0 = time.perf_counter()
        b = zl.Bases.upload(bases, expected_uses=uses)
        build_ms = (time.perf_counter() - t0) * 1e3
        try:
            c, w, l = b.plan()
            tb = b.table_bytes()
            if uses == 0:
                assert l == w and tb == l * 64 * n, (c, w, l, tb)  # one 64-byte row per window and base: 16 GiB at 16-bit windows
            else:
                assert l == 1 and tb == 64 * n, (c, w, l, tb)  # the bases themselves, no multiples
            got, ginf = b.msm(sc)
            t0 = time.perf_counter()
            got2, ginf2 = b.msm(sc)
            msm_ms = (time.perf_counter() - t0) * 1e3
            assert ginf == ginf2 == 0 and np.array_equal(got, got2)
            assert U.point_from_xy(got, ginf) == want, f"expected_uses = {uses}"
            m, off = 3 * 100352 + 17, N24 - 3 * 100352 - 17 - 5  # a sub-range ending just below the top of the key
            gs = b.msm(sc[:m], off=off, n=m)
            results[uses] = (gs, {"plan": [c, w, l], "table_bytes": tb, "upload_and_build_ms": build_ms, "msm_host_scalars_ms": msm_ms})
        finally:
            b.free()
    (s0, r0), (s1, r1) = results[0], results[1]
    assert s0[1] == s1[1] and np.array_equal(s0[0], s1[0])  # the sub-range: both handles, same bytes
    print("\nmsm_2^24:", {"table": r0, "table_less": r1})


def test_hyperkzg_setup_and_commit_at_the_largest_key(zl):
    from oracle import pymodel as pm
    from zolt_amd import api
    n, g = N_KEY, api.generator()
    tau = api.HyperKZG.TAU
    t0 = time.perf_counter()
    h, xy, inf = zl.Bases.hyperkzg_setup(g, api.fr_from_int(tau), n)
    setup_ms = (time.perf_counter() - t0) * 1e3
    h1 = None
    try:
        assert xy.shape == (n, 8) and not inf.any()
        rng = np.random.default_rng(0x2424)
        picks = [0, 1, 255, 256, 257, 65535, 65536, 65537, N24 - 1, N24, N24 + 1, n - 2, n - 1] + rng.integers(0, n, size=19).tolist()
        for i in picks:
            wxy, wi = api.MSM.scalarMul(g, api.fr_from_int(pow(tau, int(i), api.R_MOD)))
            assert wi == 0 and np.array_equal(xy[i], wxy), i
        assert U.point_from_xy(xy[n - 1], 0) == pm.ec_mul(pow(tau, n - 1, pm.R_MOD), pm.G1)  # the last power, against the Python model
        c, w, l = h.plan()
        assert l == w and h.table_bytes() == l * 64 * n
        h1, _, _ = zl.Bases.hyperkzg_setup(g, api.fr_from_int(tau), n, want_points=False, expected_uses=1)
        assert h1.plan()[2] == 1 and h1.table_bytes() == 64 * n
        # a commitment from machine words (zkvm/mod.zig:1538: evals are F.fromU64 of the trace's words): mixed magnitudes, zero runs
        words = U.splitmix64(0x574F5244, n)
        words[::3] &= np.uint64(0xFF)
        words[5::7] = 0
        words[N24 - 4096:N24] = 0
        want_w = pm.ec_mul(_horner(words.tolist(), tau, pm.R_MOD), pm.G1)
        for hh in (h, h1):
            gw = hh.msm_u64(words)
            assert U.point_from_xy(*gw) == want_w
        # and one from full-width scalars (uniform mod r)
        raw = U.random_raw256(0x4B455924, n)
        sc = zl.field_op(zl.FR, zl.OP_TO_MONT, raw)
        want_s = pm.ec_mul(_horner([v % pm.R_MOD for v in _ints256(raw)], tau, pm.R_MOD), pm.G1)
        del raw
        for hh in (h, h1):
            gs = hh.msm(sc)
            assert U.point_from_xy(*gs) == want_s
        # the part of the key a 2^24-cycle trace's commit reads when its polynomial is shorter than the key (commit takes min(len, srs))
        short = (1 << 20) + 3
        a, b = h.msm(sc[:short], n=short), h1.msm(sc[:short], n=short)
        assert a[1] == b[1] and np.array_equal(a[0], b[0])
        assert U.point_from_xy(*a) == pm.ec_mul(_horner([v % pm.R_MOD for v in _ints256(U.random_raw256(0x4B455924, n)[:short])], tau, pm.R_MOD), pm.G1)
        print("\nhyperkzg_setup_2^24+256:", {"setup_with_points_ms": setup_ms, "plan": [c, w, l], "table_bytes": h.table_bytes()})
    finally:
        h.free()
        if h1 is not None:
            h1.free()
