"""The fold kernels' narrow-factor path (fp29.hip.h FrMul): the reference's sumcheck challenges are MontU128Challenge values whose stored
Montgomery element is [0, 0, lo, hi] (tests/golden/stage2_batched_rounds.json holds 24 of them), and a fold by such a challenge takes a
9 x 5-limb product with five reduction steps instead of 9 x 9 with nine. The result must not depend on which form ran: every fold site
— the single-table session in both layouts, the product-form session's plain fold, its fused fold + evaluations and its fused
fold + expression — is held against the oracle's bind with narrow challenges, including the extreme ones."""
import hashlib

import numpy as np
import pytest

from tests import util as U

pytestmark = pytest.mark.gpu

R_TOP = 0x30644E72E131A029  # top u64 limb of r: [0, 0, lo, hi] is a canonical element for hi < R_TOP


@pytest.fixture(scope="module")
def env():
    from oracle import binding as ob
    from zolt_amd import api, lib
    lib.init()
    return api, lib, ob


def _rand(ob, seed, n):
    return ob.f_to_mont(ob.FR, U.random_raw256(seed, n))


def narrow(seed, n):
    """n stored elements [0, 0, lo, hi]: 125-bit ones like the transcript's (hi < 2^61), with the extremes in front"""
    w = U.splitmix64(seed, 2 * n).reshape(n, 2)
    out = np.zeros((n, 4), dtype=np.uint64)
    out[:, 2] = w[:, 0]
    out[:, 3] = w[:, 1] & np.uint64((1 << 61) - 1)
    edge = [(0, 0), (1, 0), (0, 1), ((1 << 64) - 1, (1 << 61) - 1), ((1 << 64) - 1, R_TOP - 1), (0, 1 << 60)]
    for i, (lo, hi) in enumerate(edge[:n]):
        out[i, 2], out[i, 3] = lo, hi
    return out


def test_narrow_elements_are_canonical(env):
    api, lib, ob = env
    for row in narrow(1, 16):
        assert int(row[0]) == 0 and int(row[1]) == 0 and sum(int(row[i]) << (64 * i) for i in range(4)) < api.R_MOD


@pytest.mark.parametrize("v,layout", [(1, 0), (2, 1), (7, 0), (8, 1), (13, 0), (16, 1), (18, 0)])
def test_session_folds_with_narrow_challenges(env, v, layout):
    api, lib, ob = env
    t = _rand(ob, 9100 + v, 1 << v)
    s = lib.SumcheckSession.open(t, lib.SC_HIGH_HALF if layout == 0 else lib.SC_LOW_PAIR)
    ch = narrow(9200 + v, v)
    cur = t
    for k in range(v):
        g0, g1 = s.round_sums()
        w0, w1 = ob.fr_sum_halves(cur) if layout == 0 else ob.fr_sum_even_odd(cur)
        assert np.array_equal(g0, w0) and np.array_equal(g1, w1), k
        s.bind(ch[k])
        cur = ob.fr_bind_high(cur, ch[k]) if layout == 0 else ob.fr_bind_low(cur, ch[k])
        assert hashlib.sha256(s.read().tobytes()).digest() == hashlib.sha256(cur.tobytes()).digest(), k
    assert np.array_equal(s.final(), cur[0])
    s.close()


def test_long_fold_narrow_and_wide_agree_with_the_oracle(env):
    """2^23 entries (the 1024-thread launch): one narrow and one full-width fold of the same table, both against the oracle"""
    api, lib, ob = env
    v = 23
    t = _rand(ob, 9300, 1 << v)
    for r in (narrow(9301, 8)[7], narrow(9301, 8)[4], _rand(ob, 9302, 1)[0]):
        s = lib.SumcheckSession.open(t, lib.SC_LOW_PAIR)
        s.bind(r)
        want = ob.fr_bind_low(t, r)
        assert hashlib.sha256(s.read().tobytes()).digest() == hashlib.sha256(want.tobytes()).digest()
        g0, g1 = s.round_sums()
        w0, w1 = ob.fr_sum_even_odd(want)
        assert np.array_equal(g0, w0) and np.array_equal(g1, w1)
        s.close()


@pytest.mark.parametrize("v", [1, 2, 6, 11, 15])
def test_product_session_plain_fold_with_narrow_challenges(env, v):
    """zg_psc_bind with no cached evaluation spec: psc_fold_kernel over every table"""
    api, lib, ob = env
    tabs = [_rand(ob, 9400 + 10 * j + v, 1 << v) for j in range(5)]
    s = lib.ProductSumcheckSession.open(tabs)
    ch = narrow(9450 + v, v)
    cur = tabs
    for k in range(v):
        s.bind(ch[k])
        cur = [ob.fr_bind_low(t, ch[k]) for t in cur]
        for j in range(5):
            assert np.array_equal(s.read(j), cur[j]), (k, j)
    s.close()


@pytest.mark.parametrize("v", [1, 3, 9, 14])
def test_prover_loops_with_narrow_challenges(env, v):
    """the fused fold + evaluations (ValEvaluation, InstructionLookups, Output) and fold + expression (InstructionInput) kernels"""
    api, lib, ob = env
    n = 1 << v
    ch = narrow(9500 + v, v)
    inc, wa, lt = (_rand(ob, 9510 + j + 10 * v, n) for j in range(3))
    claim = _rand(ob, 9520 + v, 1)[0]
    g, o = api.ValEvaluationProver(inc, wa, lt, claim), ob.ValEvaluationProver(inc, wa, lt, claim)
    for k in range(v):
        rp, wrp = g.computeRoundPolynomial(), o.computeRoundPolynomial()
        assert np.array_equal(rp, wrp), k
        g.bindChallengeWithPoly(ch[k], rp)
        o.bindChallengeWithPoly(ch[k], wrp)
        assert np.array_equal(g.current_claim, o.current_claim)
    assert all(np.array_equal(a, b) for a, b in zip(g.getFinalClaims(), o.getFinalClaims()))
    g.deinit()

    tabs = [_rand(ob, 9530 + j + 10 * v, n) for j in range(4)]
    gamma = _rand(ob, 9540 + v, 1)[0]
    g = api.InstructionLookupsClaimReductionProver(*tabs, gamma, claim)
    o = ob.InstructionLookupsClaimReduction(*tabs, gamma, claim)
    for k in range(v):
        ev, wev = g.computeRoundPolynomialCubic(), o.computeRoundPolynomialCubic()
        assert np.array_equal(ev, wev), k
        g.bindChallenge(ch[k]); o.bindChallenge(ch[k])
        g.updateClaim(ev, ch[k]); o.updateClaim(wev, ch[k])
        assert np.array_equal(g.current_claim, o.current_claim)
    fg, fo = g.getOpeningClaims(), o.getOpeningClaims()
    assert all(np.array_equal(fg[name], fo[name]) for name in fo)
    g.deinit()

    tabs = [_rand(ob, 9550 + j + 10 * v, n) for j in range(5)]
    g, o = api.OutputSumcheckProver(*tabs, claim), ob.OutputSumcheckProver(*tabs, claim)
    for k in range(v):
        ev, wev = g.roundEvals(), o.roundEvals()
        assert np.array_equal(ev, wev), k
        g.bindChallenge(ch[k]); o.bindChallenge(ch[k])
    fg, fo = g.getFinalClaims(), o.getFinalClaims()
    assert all(np.array_equal(fg[name], fo[name]) for name in fo)
    g.deinit()

    tabs = [_rand(ob, 9560 + j + 10 * v, n) for j in range(len(api.InstructionInputProver.NAMES))]
    g = api.InstructionInputProver(tabs, gamma)
    cur = [t.copy() for t in tabs]
    c = claim
    for k in range(v):
        got, want = g.computeRoundEvals(c), ob.instruction_input_round(cur, gamma, c)
        assert np.array_equal(got, want), k
        g.bind(ch[k])
        cur = [ob.fr_bind_low(t, ch[k]) for t in cur]
        c = ob.raf_update_claim(want, ch[k])
    fc = g.finalClaims()
    assert all(np.array_equal(fc[name], cur[j][0]) for j, name in enumerate(api.InstructionInputProver.NAMES))
    g.deinit()


@pytest.mark.parametrize("log_T,log_K,n", [(2, 3, 4), (8, 16, 200), (14, 32, 16384)])
def test_lasso_prover_with_narrow_challenges(env, log_T, log_K, n):
    """bit_bind_kernel's narrow form (v * r by the short product, v * (1 - r) as v - v * r) and the cycle-phase folds"""
    api, lib, ob = env
    w = _rand(ob, 9600 + log_T, log_T)
    idx = U.splitmix64(9601 + log_T, 2 * n).reshape(-1, 2).copy()
    mask = (1 << log_K) - 1
    idx[:, 0] &= np.uint64(mask & (2**64 - 1))
    idx[:, 1] &= np.uint64(mask >> 64)
    g, o = api.LassoProver(idx, log_T, log_K, w), ob.LassoProver(idx, log_T, log_K, w)
    ch = narrow(9602 + log_T, log_T + log_K)
    for rnd in range(log_T + log_K):
        assert np.array_equal(g.computeRoundPolynomial(), o.computeRoundPolynomial()), rnd
        g.receiveChallenge(ch[rnd])
        o.receiveChallenge(ch[rnd])
        assert np.array_equal(g.current_claim, o.current_claim), rnd
        if rnd % 5 == 0 or rnd >= log_K:
            assert np.array_equal(g.eq_evals(), o.eq_evals[:o.eq_evals_len]), rnd
    assert np.array_equal(g.getFinalEval(), o.getFinalEval())
    g.deinit()


@pytest.mark.parametrize("v,srs_n", [(10, 1024), (14, 16384), (17, 1 << 17)])
def test_hyperkzg_open_at_a_narrow_point(env, v, srs_n):
    """HyperKZG.open folds by the opening point's coordinates (hk_quot_fold_kernel): sumcheck challenges in the reference's flow"""
    api, lib, ob = env
    gm = ob.g1_gen_multiples(srs_n)
    inf = np.zeros(srs_n, dtype=np.uint8)
    params = api.HyperKZG.SetupParams(gm, inf)
    ev = _rand(ob, 9700 + v, 1 << v)
    pt = narrow(9710 + v, v)
    quotients, final = api.HyperKZG.open(params, ev, pt, np.zeros(4, dtype=np.uint64))
    wq, wqi, wfin = ob.hyperkzg_open(gm, inf, ev, pt, np.zeros(4, dtype=np.uint64))
    assert np.array_equal(final, wfin) and len(quotients) == v
    for i, (q, qi) in enumerate(quotients):
        assert qi == wqi[i] and np.array_equal(q, wq[i]), i
    params.deinit()


@pytest.mark.parametrize("v", [13, 14, 15, 17, 20])
def test_eq_table_with_narrow_challenges(env, v, monkeypatch):
    """The expanding eq-table kernel (csrc/poly.hip eq_expand_kernel: from 2^14 entries on, when the three variables in the middle of the
    index are 128-bit challenges, a thread forms one full product and expands it as the reference's own doubling build does — seven 9 x 5-limb
    products and seven subtractions for eight entries) against the oracle's evalsSliceWithScaling, with every challenge narrow (extremes
    included: 0, 1, the largest canonical [0, 0, lo, hi]), with only the middle three narrow, with one of them wide (the one-product-per-
    entry kernel must take over), with a scaling factor, and with the kernel switched off: identical bytes every time."""
    api, lib, ob = env
    wide = _rand(ob, 4400 + v, v)
    nar = narrow(4500 + v, v)
    xh = v - 11
    mid = wide.copy()
    if v >= 14:
        mid[xh:xh + 3] = nar[:3]  # the three extremes 0, 1 * 2^128, 2^192 sit exactly on the expansion variables
    one_wide = nar.copy()
    if v >= 14:
        one_wide[xh + 1] = wide[0]
    scale = _rand(ob, 4600 + v, 1)[0]
    for r, sc in ((nar, None), (mid, None), (one_wide, scale), (nar[::-1].copy(), scale), (wide, None)):
        want = ob.fr_eq_table(r, sc)
        got = lib.fr_eq_table(r, sc)
        assert hashlib.sha256(got.tobytes()).digest() == hashlib.sha256(want.tobytes()).digest()


def test_eq_table_expanding_kernel_at_every_size_in_a_child_process(tmp_path):
    """the expanding kernel is the default from 2^20 entries on; ZG_EQ_EXPAND = 14 (read once per process) forces it from 2^14 on, so that its
    row / workgroup geometry is exercised at every size between: v = 14 .. 19 against the oracle, in a fresh process"""
    import os
    import subprocess
    import sys
    code = (
        "import hashlib, numpy as np\n"
        "from oracle import binding as ob\n"
        "from zolt_amd import lib\n"
        "from tests.test_gpu_narrow_challenges import narrow\n"
        "lib.init(0)\n"
        "for v in range(14, 20):\n"
        "    r = narrow(9000 + v, v)[::-1].copy()\n"
        "    assert hashlib.sha256(lib.fr_eq_table(r).tobytes()).digest() == hashlib.sha256(ob.fr_eq_table(r).tobytes()).digest(), v\n"
        "print('ok')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ZG_EQ_EXPAND="14", PYTHONPATH=root), cwd=root, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "ok" in res.stdout, res.stdout[-1000:] + res.stderr[-2000:]
