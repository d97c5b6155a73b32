"""eq-table / fold / sumcheck parity: HIP kernels vs the CPU oracle through the C ABI."""
import hashlib
import json
import os

import numpy as np
import pytest

from tests import util as U

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def zl():
    from zolt_amd import lib
    lib.init()
    return lib


@pytest.fixture(scope="module")
def ob():
    from oracle import binding
    return binding


def _rand(ob, seed, n):
    return ob.f_to_mont(ob.FR, U.random_raw256(seed, n))


@pytest.mark.parametrize("v", [0, 1, 2, 3, 7, 8, 9, 13, 16])
def test_eq_table_vs_oracle(zl, ob, v):
    r = _rand(ob, 300 + v, v)
    assert np.array_equal(zl.fr_eq_table(r), ob.fr_eq_table(r))
    scale = _rand(ob, 400 + v, 1)[0]
    assert np.array_equal(zl.fr_eq_table(r, scale), ob.fr_eq_table(r, scale))


def test_eq_table_golden(zl, ob):
    vec = U.load_vectors()
    for case in vec["eq_table"]:
        r = U.fr_hex(case["r"]) if case["r"] else np.zeros((0, 4), dtype=np.uint64)
        got = zl.fr_eq_table(r)
        assert [U.fr_to_int(x) for x in got] == [int(h, 16) for h in case["table"]]
    # the captured run's 13 Stage-1 tau challenges (reference logs/zolt.log:47-59)
    taus = json.load(open(os.path.join(U.GOLDEN, "stage1_tau.json")))["tau_hex"]
    from oracle import pymodel as pm
    r = U.fr([int(h, 16) % pm.R_MOD for h in taus])
    got = zl.fr_eq_table(r)
    assert np.array_equal(got, ob.fr_eq_table(r))
    assert np.array_equal(got, ob.fr_eq_table_append_lsb(r))  # GruenSplitEq build order, same table


def test_eq_table_matches_reference_log(zl, ob):
    """zg_fr_eq_table against the E_out / E_in entries the reference printed for a real GruenSplitEqPolynomial
    (logs/zolt.log [STAGE4_GRUEN_INIT]; fixture tests/golden/stage4_gruen_eq.json, src/poly/split_eq.zig:91-171)."""
    d = json.load(open(os.path.join(U.GOLDEN, "stage4_gruen_eq.json")))
    tau = np.array([[int(x) for x in row] for row in d["r_cycle_be_mont_limbs"]], dtype=np.uint64)
    n, m = d["n"], d["m"]
    for name, sl, ln in (("E_out", slice(0, m), d["E_out_len"]), ("E_in", slice(m, n - 1), d["E_in_len"])):
        t = zl.fr_eq_table(tau[sl])
        assert len(t) == ln
        tc = zl.field_op(zl.FR, zl.OP_FROM_MONT, t)  # canonical limbs (F.toBytes = LE bytes of these)
        assert [tc[i].tobytes().hex() for i in range(4)] == d[name + "_first4_canonical_le_hex"], name
        assert np.array_equal(t, ob.fr_eq_table(tau[sl]))


@pytest.mark.parametrize("v", [0, 1, 2, 5, 8, 9, 12, 15])
def test_eq_prefix_tables_vs_oracle(zl, ob, v):
    """zg_fr_eq_prefix_tables[_dev]: GruenSplitEqPolynomial's E_vec (src/poly/split_eq.zig:122-171), every level, one launch"""
    tau = _rand(ob, 900 + v, v)
    got, want = zl.fr_eq_prefix_tables(tau), ob.fr_eq_prefix_tables(tau)
    assert len(got) == len(want) == v + 1
    for k in range(v + 1):
        assert got[k].shape == (1 << k, 4) and np.array_equal(got[k], want[k]), k
    buf = zl.DeviceBuffer(((2 << v) - 1) * 32)
    zl.fr_eq_prefix_tables_dev(tau, buf.ptr)
    flat = buf.to_host(np.uint64).reshape(-1, 4)
    assert np.array_equal(flat, np.concatenate(want))
    buf.free()


def test_gruen_split_eq_mirror_reference_inline_tests(zl, ob):
    """src/poly/split_eq.zig:525-733 restated against zolt_amd.api.GruenSplitEqPolynomial (tables from the device)."""
    from zolt_amd import api
    F = api.fr_from_int
    P = api.R_MOD
    p = api.GruenSplitEqPolynomial.init(np.stack([F(2), F(3), F(5)]))
    assert p.current_index == 3 and np.array_equal(p.current_scalar, F(1))
    assert (p.num_x_in, p.num_x_out, len(p.E_in_vec), len(p.E_out_vec)) == (1, 1, 2, 2)
    assert np.array_equal(p.E_out_vec[1], np.stack([F(P - 1), F(2)])) and np.array_equal(p.E_in_vec[1], np.stack([F(P - 2), F(3)]))
    p = api.GruenSplitEqPolynomial.init(np.stack([F(2), F(3)]))
    p.bind(F(5))
    assert p.current_index == 1 and np.array_equal(p.current_scalar, F(23))
    p = api.GruenSplitEqPolynomial.init(np.stack([F(1), F(2)]))
    rp = p.computeCubicRoundPoly(F(10), F(3), F(100))
    assert (api.fr_to_int(rp[0]) + api.fr_to_int(rp[1])) % P == 100
    p = api.GruenSplitEqPolynomial.init(np.stack([F(3), F(5), F(7), F(11)]))
    t = [api.fr_to_int(x) for x in p.getFullEqTable()]
    assert len(t) == 16 and t[0] == (-2 * -4 * -6 * -10) % P and t[15] == 3 * 5 * 7 * 11
    assert t[5] == (-2 * 5 * -6 * 11) % P and t[10] == (3 * -4 * 7 * -10) % P
    assert [api.fr_to_int(x) for x in p.getEActiveForWindow(1)] == [1]
    assert [api.fr_to_int(x) for x in p.getEActiveForWindow(2)] == [(-6) % P, 7]
    assert [api.fr_to_int(x) for x in p.getEActiveForWindow(3)] == [24, (-28) % P, (-30) % P, 35]
    assert [api.fr_to_int(x) for x in p.getEActiveForWindow(5)] == [1]  # wider than the unbound variables (:476-481)


def test_gruen_mirror_bind_of_the_captured_run(zl, ob):
    """api.GruenSplitEqPolynomial against the split_eq scalars and window sizes the reference printed for its ProductVirtualRemainder
    instance (fixture stage2_batched_rounds.json; see tests/test_transcript_host.py::test_gruen_split_eq_bind_of_the_captured_run)"""
    from zolt_amd import api
    d = json.load(open(os.path.join(U.GOLDEN, "stage2_batched_rounds.json")))
    pr = d["product_remainder"]
    M = lambda h: api.fr_from_int(int.from_bytes(bytes.fromhex(h), "little"))
    tau = np.stack([M(h) for h in d["stage1_r_cycle"]])
    g = api.GruenSplitEqPolynomial.initWithScaling(tau, M(pr["current_scalar_before_round"][0]))
    for k in range(3):
        e_out, e_in, _ = g.getWindowEqTables(0, 1)
        assert (len(e_out), len(e_in)) == (pr["E_out_len"][k], pr["E_in_len"][k])
        d_out, n_out, d_in, n_in = g.getWindowEqTablesDev(1)
        assert (n_out, n_in) == (pr["E_out_len"][k], pr["E_in_len"][k])
        assert np.array_equal(g.current_scalar, M(pr["current_scalar_before_round"][k]))
        g.bind(M(d["rounds"][pr["first_batch_round"] + k]["challenge"]))
    g.deinit()
    # the eq table the same run built over these eight challenges for its opening claims: first three of 256 entries are in the log
    eq = zl.fr_eq_table(tau)
    assert len(eq) == 256 and all(np.array_equal(eq[i], M(h)) for i, h in enumerate(d["eq_evals_of_r_cycle_first3"]))
    # and computeOpeningClaims' table over the reversed Stage-2 cycle challenges ("FACTOR_EVALS: eq_evals[k]")
    first = pr["first_batch_round"]
    r2 = np.stack([M(d["rounds"][k]["challenge"]) for k in range(len(d["rounds"]) - 1, first - 1, -1)])
    eq2 = zl.fr_eq_table(r2)
    assert all(np.array_equal(eq2[i], M(h)) for i, h in enumerate(d["eq_evals_of_reversed_stage2_challenges_first3"]))


@pytest.mark.parametrize("n", [0, 1, 2, 7, 8, 13, 24])
def test_gruen_split_eq_mirror_vs_oracle_through_all_rounds(zl, ob, n):
    """The mirror against the oracle's restatement of the struct across a whole LowToHigh binding: tables, window views,
    scalar, cubic round polynomial, active-window and full tables at every round; n = 8 is the captured run's shape (fixture)."""
    from zolt_amd import api
    if n == 8:
        d = json.load(open(os.path.join(U.GOLDEN, "stage4_gruen_eq.json")))
        tau = np.array([[int(x) for x in row] for row in d["r_cycle_be_mont_limbs"]], dtype=np.uint64)
    else:
        tau = _rand(ob, 950 + n, n)
    scale = _rand(ob, 980 + n, 1)[0] if n % 2 else None
    g, w = api.GruenSplitEqPolynomial(tau, scale), ob.GruenSplitEq(tau, scale)
    rs = _rand(ob, 990 + n, 3 * n + 3)
    for rnd in range(n + 1):
        assert g.current_index == w.current_index and np.array_equal(g.current_scalar, w.current_scalar)
        assert len(g.E_out_vec) == len(w.E_out_vec) and len(g.E_in_vec) == len(w.E_in_vec)
        for a, b in zip(g.E_out_vec + g.E_in_vec, w.E_out_vec + w.E_in_vec):
            assert np.array_equal(a, b)
        if n:
            for ws in (1, 2, 3):
                eo, ei, hib = g.getWindowEqTables(0, ws)
                weo, wei, whib = w.getWindowEqTables(ws)
                assert hib == whib and np.array_equal(eo, weo) and np.array_equal(ei, wei)
                assert np.array_equal(g.getEActiveForWindow(ws), w.getEActiveForWindow(ws))
        q0, q2, claim = rs[3 * rnd], rs[3 * rnd + 1], rs[3 * rnd + 2]
        assert np.array_equal(g.computeCubicRoundPoly(q0, q2, claim), w.computeCubicRoundPoly(q0, q2, claim))
        if g.current_index <= 16:
            assert np.array_equal(g.getFullEqTable(), w.getFullEqTable())
        assert np.array_equal(g.getTauHigh(), w.getTauHigh())
        g.bind(rs[rnd])
        w.bind(rs[rnd])


@pytest.mark.parametrize("T,k", [(1, 36), (2, 36), (3, 36), (256, 36), (1000, 36), (4096, 8), (1 << 16, 64), (1 << 14, 1)])
def test_r1cs_claimed_inputs_vs_oracle(zl, ob, T, k):
    """R1CSInputEvaluator.computeClaimedInputs (src/zkvm/r1cs/evaluation.zig:55-122): k column MLEs at r_cycle from the cycle-major
    witness matrix; a cycle count that is not a power of two uses the floor power of two, as the reference's log2_int does"""
    from zolt_amd import api
    w = _rand(ob, 1300 + k, T * k).reshape(T, k, 4)
    w[U.splitmix64(1301, T * k).reshape(T, k) % np.uint64(4) == 0] = 0
    log_n = T.bit_length() - 1
    r = _rand(ob, 1302 + T % 89, log_n)
    got = api.R1CSInputEvaluator.computeClaimedInputs(w, r)
    assert np.array_equal(got, ob.r1cs_claimed_inputs(w, r))
    assert np.array_equal(api.R1CSInputEvaluator.computeClaimedInput(w, r, k - 1), got[k - 1])
    if log_n:
        longer = np.concatenate([r, _rand(ob, 1303, 2)])  # only the first log_n challenges are used (:71-73)
        assert np.array_equal(api.R1CSInputEvaluator.computeClaimedInputs(w, longer), got)
        assert np.array_equal(api.R1CSInputEvaluator.computeClaimedInputs(w, r[:0]), w[0])  # no cycle variables: the first witness (:75-83)
        assert np.array_equal(ob.r1cs_claimed_inputs(w, r[:0]), w[0])
    if log_n > 1:  # fewer challenges than variables: the reference would index eq_evals out of bounds
        with pytest.raises(IndexError):
            api.R1CSInputEvaluator.computeClaimedInputs(w, r[:-1])
        with pytest.raises(IndexError):
            ob.r1cs_claimed_inputs(w, r[:-1])


def test_rows_mle_direct(zl, ob):
    """zg_fr_rows_mle[_dev]: fewer rows than the hypercube (the missing rows count as zero), rows beyond it ignored, argument checks"""
    T, k, v = 300, 5, 9
    w = _rand(ob, 1310, T * k).reshape(T, k, 4)
    r = _rand(ob, 1311, v)
    eq = ob.fr_eq_table(r)
    want = np.stack([ob.fr_sum_halves(np.concatenate([ob.f_mul(ob.FR, eq[:T], w[:, i]), np.zeros((2 * 512 - T, 4), dtype=np.uint64)]))[0] for i in range(k)])
    assert np.array_equal(zl.fr_rows_mle(w, r), want)
    d = zl.DeviceBuffer.from_host(w)
    assert np.array_equal(zl.fr_rows_mle_dev(d.ptr, T, k, r), want)
    assert np.array_equal(zl.fr_rows_mle_dev(d.ptr, T, k, r[:0]), w[0])  # v = 0: eq = [1], one row
    big = _rand(ob, 1312, 20 * k).reshape(20, k, 4)
    assert np.array_equal(zl.fr_rows_mle(big, r[:3]), zl.fr_rows_mle(big[:8], r[:3]))
    d.free()
    with pytest.raises(RuntimeError):
        zl.fr_rows_mle(_rand(ob, 1313, 65 * 2).reshape(2, 65, 4), r[:1])


@pytest.mark.parametrize("v", [0, 1, 2, 5, 9, 11])
def test_eq_plus_one_table_vs_oracle(zl, ob, v):
    """zg_fr_eq_plus_one_table[_dev] against the oracle evaluating EqPlusOnePolynomial.mle at every cube point (src/poly/mod.zig:407-435,
    530-548); EqPlusOnePrefixSuffixPoly's four tables (:462-528)"""
    from zolt_amd import api
    r = _rand(ob, 1400 + v, v)
    want = ob.eq_plus_one_table(r)
    assert np.array_equal(zl.fr_eq_plus_one_table(r), want)
    assert np.array_equal(api.EqPlusOnePolynomial(r).evals(), want)
    buf = zl.DeviceBuffer((1 << v) * 32)
    zl.fr_eq_plus_one_table_dev(r, buf.ptr)
    assert np.array_equal(buf.to_host().reshape(-1, 4), want)
    buf.free()
    if v >= 2:
        ps = api.EqPlusOnePrefixSuffixPoly(r)
        mid = v // 2
        assert np.array_equal(ps.prefix_0, ob.eq_plus_one_table(r[mid:])) and np.array_equal(ps.suffix_0, ob.fr_eq_table(r[:mid]))
        assert np.array_equal(ps.suffix_1, ob.eq_plus_one_table(r[:mid])) and ps.prefixSize() == 1 << (v - mid) and ps.suffixSize() == 1 << mid
        ones = np.tile(ob.f_from_u64(ob.FR, np.array([1], dtype=np.uint64)), (v - mid, 1))
        assert np.array_equal(ps.prefix_1[0], ob.fr_eq_mle(ones, r[mid:])) and not ps.prefix_1[1:].any()


def test_bind_kats_and_golden(zl, ob):
    """src/poly/mod.zig:816-888 and tests/golden folds."""
    got = zl.fr_bind_low(U.fr([1, 2, 3, 4]), U.fr([3])[0])
    assert [U.fr_to_int(x) for x in got] == [4, 6]
    got = zl.fr_bind_low(U.fr([10, 20, 30, 40, 50, 60, 70, 80]), U.fr([5])[0])
    assert [U.fr_to_int(x) for x in got] == [60, 80, 100, 120]
    got = zl.fr_bind_high(U.fr([1, 2, 3, 4]), U.fr([2])[0])  # src/subprotocols/mod.zig:426-432
    assert [U.fr_to_int(x) for x in got] == [5, 6]
    for case in U.load_vectors()["folds"]:
        t, r = U.fr_hex(case["table"]), U.fr_hex([case["r"]])[0]
        assert [U.fr_to_int(x) for x in zl.fr_bind_high(t, r)] == [int(h, 16) for h in case["bind_high"]]
        assert [U.fr_to_int(x) for x in zl.fr_bind_low(t, r)] == [int(h, 16) for h in case["bind_low"]]


@pytest.mark.parametrize("logn", [1, 2, 5, 9, 14, 18])
def test_binds_vs_oracle(zl, ob, logn):
    t = _rand(ob, 500 + logn, 1 << logn)
    r = _rand(ob, 600 + logn, 1)[0]
    assert np.array_equal(zl.fr_bind_high(t, r), ob.fr_bind_high(t, r))
    assert np.array_equal(zl.fr_bind_low(t, r), ob.fr_bind_low(t, r))
    assert np.array_equal(zl.fr_bind_low(t, r), ob.fr_bind_low_2mul(t, r))  # jolt_r1cs.zig:470-477 form


def test_spartan_combine(zl, ob):
    n = 5000
    eq, az, bz, cz = (_rand(ob, 700 + k, n) for k in range(4))
    cz[n // 2:] = 0  # zero-padded past num_constraints (src/zkvm/spartan/mod.zig:191-199)
    assert np.array_equal(zl.fr_spartan_combine(eq, az, bz, cz), ob.fr_spartan_combine(eq, az, bz, cz))


def _run_sumcheck_gpu(zl, ob, evals):
    """runSumcheck (src/subprotocols/mod.zig:302-354) with the table resident on the GPU; the toy
    verifier's challenge derivation stays on the host exactly like the reference."""
    from oracle import pymodel as pm
    s = zl.SumcheckSession.open(evals, zl.SC_HIGH_HALF)
    g0, g1 = s.round_sums()
    claim = ob.f_add(ob.FR, g0, g1)
    vclaim = claim
    rounds, chals = [], []
    rd = 0
    while len(s) > 1:
        g0, g1 = s.round_sums()
        coeffs = np.stack([g0, ob.f_sub(ob.FR, g1, g0)])
        assert np.array_equal(ob.f_add(ob.FR, ob.f_add(ob.FR, coeffs[0], coeffs[0]), coeffs[1]), vclaim)
        ch = ob.sumcheck_derive_challenge(rd, vclaim, coeffs)
        vclaim = ob.f_add(ob.FR, ob.f_mul(ob.FR, coeffs[1], ch), coeffs[0])
        s.bind(ch)
        rounds.append(coeffs)
        chals.append(ch)
        rd += 1
    fin = s.final()
    s.close()
    return claim, np.array(rounds), np.array(chals), fin, int(np.array_equal(vclaim, fin))


@pytest.mark.parametrize("logn", [1, 3, 6, 12, 16])
def test_run_sumcheck_vs_oracle(zl, ob, logn):
    evals = _rand(ob, 800 + logn, 1 << logn)
    c, rds, chs, fin, ok = _run_sumcheck_gpu(zl, ob, evals)
    wc, wr, wch, wfin, wok = ob.run_sumcheck(evals)
    assert np.array_equal(c, wc) and np.array_equal(rds, wr) and np.array_equal(chs, wch)
    assert np.array_equal(fin, wfin) and ok == wok == 1


@pytest.mark.parametrize("logn", [0, 1, 2, 3, 6, 9, 10, 12, 13, 16, 17, 18])
def test_device_resident_run_sumcheck_vs_oracle(zl, ob, logn):
    """zg_run_sumcheck: prover and toy verifier both on the device (src/subprotocols/mod.zig:302-354, :165-243) —
    claim, every round polynomial, every challenge, the final evaluation and the result flag equal the oracle's."""
    evals = _rand(ob, 1800 + logn, 1 << logn)
    res = zl.run_sumcheck(evals)
    wc, wr, wch, wfin, wok = ob.run_sumcheck(evals)
    assert np.array_equal(res["claim"], wc) and np.array_equal(res["final_eval"], wfin) and res["result"] == bool(wok)
    assert np.array_equal(res["rounds"].reshape(-1, 2, 4), np.asarray(wr).reshape(-1, 2, 4))
    assert np.array_equal(res["final_point"].reshape(-1, 4), np.asarray(wch).reshape(-1, 4))
    # the input table is left untouched by the device path; the other launch shapes (every round its own launch;
    # the LDS-resident tail taking over at 64 entries) give the same transcript
    import os
    for tail in ("1", "64"):
        os.environ["ZG_SC_TAIL_MAX"] = tail
        try:
            again = zl.run_sumcheck(evals)
        finally:
            del os.environ["ZG_SC_TAIL_MAX"]
        assert all(np.array_equal(res[k], again[k]) for k in ("claim", "rounds", "final_point", "final_eval")) and again["result"]


def test_device_resident_run_sumcheck_kats_and_full_size(zl, ob):
    res = zl.run_sumcheck(U.fr(range(1, 9)))  # src/subprotocols/mod.zig:441-461
    assert U.fr_to_int(res["claim"]) == 36 and res["result"] and res["rounds"].shape == (3, 2, 4)
    for case in U.load_vectors()["sumcheck"]:
        res = zl.run_sumcheck(U.fr_hex(case["evals"]))
        assert U.fr_to_int(res["claim"]) == int(case["claim"], 16)
        assert [[U.fr_to_int(x) for x in rd] for rd in res["rounds"]] == [[int(h, 16) for h in rd] for rd in case["rounds"]]
        assert U.fr_to_int(res["final_eval"]) == int(case["final_eval"], 16) and res["result"] == bool(int(case["ok"]))
    evals = _rand(ob, 0x53554D43, 1 << 20)  # BASELINE config 3 size
    res = zl.run_sumcheck(evals)
    wc, wr, wch, wfin, wok = ob.run_sumcheck(evals)
    assert res["result"] and wok == 1 and np.array_equal(res["claim"], wc) and np.array_equal(res["final_eval"], wfin)
    assert np.array_equal(res["rounds"].reshape(-1, 2, 4), np.asarray(wr).reshape(-1, 2, 4))
    assert np.array_equal(res["final_point"].reshape(-1, 4), np.asarray(wch).reshape(-1, 4))
    with pytest.raises(zl.ZgError):
        zl.run_sumcheck(evals[:100])  # length must be a power of two


@pytest.mark.parametrize("v,layout", [(0, 0), (1, 0), (3, 1), (8, 0), (9, 1), (12, 0), (14, 1), (16, 0)])
def test_session_opened_from_spartan_inputs(zl, ob, v, layout):
    """zg_sumcheck_open_spartan_dev: f = eq(r,.)*(Az*Bz - Cz) built straight into the session with round 0's sums
    (src/zkvm/spartan/mod.zig:182-206 + Sumcheck.Prover.init) equals eq table -> combine -> open -> round_sums done separately
    with the oracle, including the scaled eq table, both fold layouts, and the rounds that follow."""
    import ctypes as C
    n = 1 << v
    r = _rand(ob, 3000 + v, v) if v else np.zeros((0, 4), dtype=np.uint64)
    scale = _rand(ob, 3100 + v, 1)[0] if v % 2 else None
    az, bz, cz = (_rand(ob, 3200 + 3 * v + k, n) for k in range(3))
    want = ob.fr_spartan_combine(ob.fr_eq_table(r, scale), az, bz, cz)
    d = []
    for t in (az, bz, cz):
        p = C.c_void_p()
        assert zl._lib.zg_dev_alloc(C.c_size_t(n * 32), C.byref(p)) == 0
        assert zl._lib.zg_memcpy_h2d(p, np.ascontiguousarray(t).ctypes.data_as(C.c_void_p), C.c_size_t(n * 32)) == 0
        d.append(p)
    s = zl.SumcheckSession.open_spartan_dev(r, d[0].value, d[1].value, d[2].value, layout=layout, scale=scale)
    assert len(s) == n and np.array_equal(s.read(), want)
    cur = want
    chals = _rand(ob, 3300 + v, max(v, 1))
    for k in range(v):
        g0, g1 = s.round_sums()
        w0, w1 = ob.fr_sum_halves(cur) if layout == 0 else ob.fr_sum_even_odd(cur)
        assert np.array_equal(g0, w0) and np.array_equal(g1, w1), k
        s.bind(chals[k])
        cur = ob.fr_bind_high(cur, chals[k]) if layout == 0 else ob.fr_bind_low(cur, chals[k])
    assert np.array_equal(s.final(), cur[0])
    s.close()
    for p in d:
        zl._lib.zg_dev_free(p)


def test_sumcheck_kats(zl, ob):
    """src/subprotocols/mod.zig:366-461: [1,2,3,4] -> g(0)=3, g(1)=7, r=2 -> [5,6]; [1..8] -> claim 36."""
    s = zl.SumcheckSession.open(U.fr([1, 2, 3, 4]))
    g0, g1 = s.round_sums()
    assert (U.fr_to_int(g0), U.fr_to_int(g1)) == (3, 7)
    s.bind(U.fr([2])[0])
    assert [U.fr_to_int(x) for x in s.read()] == [5, 6]
    g0, g1 = s.round_sums()
    assert (U.fr_to_int(g0), U.fr_to_int(g1)) == (5, 6)
    s.close()
    c, rds, chs, fin, ok = _run_sumcheck_gpu(zl, ob, U.fr(range(1, 9)))
    assert U.fr_to_int(c) == 36 and ok == 1
    for case in U.load_vectors()["sumcheck"]:
        c, rds, chs, fin, ok = _run_sumcheck_gpu(zl, ob, U.fr_hex(case["evals"]))
        assert U.fr_to_int(c) == int(case["claim"], 16)
        assert [[U.fr_to_int(x) for x in rd] for rd in rds] == [[int(h, 16) for h in rd] for rd in case["rounds"]]
        assert U.fr_to_int(fin) == int(case["final_eval"], 16) and ok == int(case["ok"])


def test_low_pair_session_every_table(zl, ob):
    """BASELINE config 3 mode (ii) at a CPU-checkable size: eq-table -> combine -> LowToHigh folds with given
    challenges; SHA-256 of every intermediate table and every round's (even, odd) sums equal the oracle's."""
    v = 14
    r = _rand(ob, 900, v)
    eq = zl.fr_eq_table(r)
    az, bz, cz = (_rand(ob, 901 + k, 1 << v) for k in range(3))
    f_gpu = zl.fr_spartan_combine(eq, az, bz, cz)
    f_cpu = ob.fr_spartan_combine(ob.fr_eq_table(r), az, bz, cz)
    assert np.array_equal(f_gpu, f_cpu)
    chals = _rand(ob, 905, v)
    s = zl.SumcheckSession.open(f_gpu, zl.SC_LOW_PAIR)
    cur = f_cpu
    for k in range(v):
        g0, g1 = s.round_sums()
        w0, w1 = ob.fr_sum_even_odd(cur)
        assert np.array_equal(g0, w0) and np.array_equal(g1, w1), k
        s.bind(chals[k])
        cur = ob.fr_bind_low(cur, chals[k])
        assert hashlib.sha256(s.read().tobytes()).digest() == hashlib.sha256(cur.tobytes()).digest(), k
    assert np.array_equal(s.final(), cur[0])
    s.close()


@pytest.mark.parametrize("layout_name", ["SC_LOW_PAIR", "SC_HIGH_HALF"])
@pytest.mark.parametrize("v", [0, 1, 2, 9, 14])
def test_session_on_a_borrowed_table(zl, ob, v, layout_name):
    """zg_sumcheck_open_dev_borrowed: the session reads the caller's device table in place (no copy) until its first bind, then folds in
    buffers of its own — every round's sums, every table after a bind and the final value equal the copying session's and the oracle's;
    the caller's table is left as it was; an in-place bit-bind before the first bind is refused; a closed borrowed session goes back to the
    pool and serves an ordinary one."""
    layout = getattr(zl, layout_name)
    n = 1 << v
    tab = _rand(ob, 1700 + v, n)
    chals = _rand(ob, 1750 + v, max(v, 1))
    d = zl.DeviceBuffer.from_host(tab)
    for rep in range(2):  # the second pass takes the first one's session from the pool
        s = zl.SumcheckSession.open_dev(d.ptr, n, layout, borrow=True)
        c = zl.SumcheckSession.open(tab, layout)
        cur = tab
        assert np.array_equal(s.read(), tab)
        for k in range(v):
            assert all(np.array_equal(a, b) for a, b in zip(s.round_sums(), c.round_sums())), k
            want = ob.fr_sum_even_odd(cur) if layout == zl.SC_LOW_PAIR else ob.fr_sum_halves(cur)
            assert all(np.array_equal(a, b) for a, b in zip(s.round_sums(), want)), k
            s.bind(chals[k])
            c.bind(chals[k])
            cur = ob.fr_bind_low(cur, chals[k]) if layout == zl.SC_LOW_PAIR else ob.fr_bind_high(cur, chals[k])
            assert np.array_equal(s.read().reshape(-1, 4), cur), k
        assert np.array_equal(s.final(), cur[0]) and np.array_equal(c.final(), cur[0])
        s.close()
        c.close()
        assert np.array_equal(d.to_host().reshape(-1, 4), tab)
    if v >= 2:
        s = zl.SumcheckSession.open_dev(d.ptr, n, zl.SC_HIGH_HALF, borrow=True)
        idx = zl.DeviceBuffer.from_host(np.zeros((n, 2), dtype=np.uint64))
        with pytest.raises(zl.ZgError):
            s.bit_bind(idx.ptr, n, 0, chals[0])
        s.close()
        idx.free()
    d.free()


def test_full_size_low_pair_session_from_spartan_inputs(zl, ob):
    """BASELINE config 3 mode (ii) at FULL size (v = 20): the fused eq x (Az*Bz - Cz) open (zg_sumcheck_open_spartan_dev:
    src/zkvm/spartan/mod.zig:182-206 + Sumcheck.Prover.init), then 20 LowToHigh rounds with given challenges (the reference's 128-bit
    challenge shape on odd rounds, full-width elements on even ones: both multiplier forms). Every round's (even, odd) pair equals the
    oracle's; SHA-256 of the whole table equals the oracle's after rounds 0, 1, 2, 10 and at the end (the oracle folds on one core:
    ~1 s for the first fold)."""
    import ctypes as C
    v = 20
    n = 1 << v
    r = _rand(ob, 0x4C4F5750, v)
    az, bz, cz = (_rand(ob, 0x4C4F5751 + k, n) for k in range(3))
    cur = ob.fr_spartan_combine(ob.fr_eq_table(r), az, bz, cz)
    d = []
    for t in (az, bz, cz):
        p = C.c_void_p()
        assert zl._lib.zg_dev_alloc(C.c_size_t(n * 32), C.byref(p)) == 0
        assert zl._lib.zg_memcpy_h2d(p, np.ascontiguousarray(t).ctypes.data_as(C.c_void_p), C.c_size_t(n * 32)) == 0
        d.append(p)
    s = zl.SumcheckSession.open_spartan_dev(r, d[0].value, d[1].value, d[2].value, layout=zl.SC_LOW_PAIR)
    assert len(s) == n
    assert hashlib.sha256(s.read().tobytes()).digest() == hashlib.sha256(cur.tobytes()).digest(), "table after the fused open"
    chals = _rand(ob, 0x4C4F5760, v)
    chals[1::2, :2] = 0  # MontU128Challenge: the stored Montgomery element is [0, 0, lo, hi]
    for k in range(v):
        g0, g1 = s.round_sums()
        w0, w1 = ob.fr_sum_even_odd(cur)
        assert np.array_equal(g0, w0) and np.array_equal(g1, w1), k
        s.bind(chals[k])
        cur = ob.fr_bind_low(cur, chals[k])
        if k in (0, 1, 2, 10, v - 1):
            assert hashlib.sha256(s.read().tobytes()).digest() == hashlib.sha256(cur.tobytes()).digest(), k
    assert np.array_equal(s.final(), cur[0])
    s.close()
    for p in d:
        zl._lib.zg_dev_free(p)


@pytest.mark.parametrize("logn", [6, 12, 20, 22])
def test_lazy_sums_at_the_largest_stored_values(zl, ob, logn):
    """The round sums are kept as plain 288-bit integers and reduced once per workgroup (csrc/sc_common.hip.h: Acc9, acc9_reduce's
    quotient estimate from the top 64 bits). Worst case for the estimate and the carries: every stored element is r - 1 (the
    largest canonical limb pattern), so a half-table of 2^(logn-1) entries sums to 2^(logn-1) (r - 1) — top limb and quotient as
    large as that length allows. Closed form in Python integers; both layouts; then a fold by r - 1 and the next sums."""
    R = U.pm.R_MOD if hasattr(U, "pm") else 21888242871839275222246405745257275088548364400416034343698204186575808495617
    n = 1 << logn
    top = np.array([(R - 1) >> (64 * i) & ((1 << 64) - 1) for i in range(4)], dtype=np.uint64)
    tab = np.repeat(top[None, :], n, axis=0)
    want = (n // 2) * (R - 1) % R
    limbs = lambda v: np.array([(v >> (64 * i)) & ((1 << 64) - 1) for i in range(4)], dtype=np.uint64)
    for layout in (zl.SC_HIGH_HALF, zl.SC_LOW_PAIR):
        s = zl.SumcheckSession.open(tab, layout)
        g0, g1 = s.round_sums()
        assert np.array_equal(g0, limbs(want)) and np.array_equal(g1, limbs(want)), (logn, layout)
        if logn <= 12:  # the oracle folds on one core: keep the fold check to the short tables
            s.bind(top)
            cur = ob.fr_bind_high(tab, top) if layout == zl.SC_HIGH_HALF else ob.fr_bind_low(tab, top)
            w0, w1 = ob.fr_sum_halves(cur) if layout == zl.SC_HIGH_HALF else ob.fr_sum_even_odd(cur)
            g0, g1 = s.round_sums()
            assert np.array_equal(g0, w0) and np.array_equal(g1, w1)
        s.close()


def test_full_size_sumcheck_properties(zl, ob):
    """BASELINE config 3 at full size (v = 20): size-independent checks — every round satisfies
    g0 + g1 == previous claim evaluated at the challenge, and the final evaluation equals the
    multilinear extension evaluated at the challenge point (computed on the GPU by an eq-table inner
    product is avoided: we use the verifier identity instead), plus oracle equality on the round messages."""
    v = 20
    evals = _rand(ob, 0x53554D43, 1 << v)
    c, rds, chs, fin, ok = _run_sumcheck_gpu(zl, ob, evals)
    assert ok == 1
    wc, wr, wch, wfin, wok = ob.run_sumcheck(evals)  # ~1 s on the CPU
    assert np.array_equal(c, wc) and np.array_equal(rds, wr) and np.array_equal(chs, wch) and np.array_equal(fin, wfin)
    # eq table at v = 20: partition of unity and spot rows against the oracle
    r = _rand(ob, 0x45515F54, v)
    got = zl.fr_eq_table(r)
    want = ob.fr_eq_table(r)
    assert hashlib.sha256(got.tobytes()).digest() == hashlib.sha256(want.tobytes()).digest()


def test_dense_polynomial_add_scale_and_eq_mle(zl, ob):
    """DensePolynomial.add / scale (src/poly/mod.zig:94-126) and EqPolynomial.mle / evaluate (:214-227,311-321) through the host
    mirror, against the oracle. mle at a boolean point is the eq table's entry (big-endian index)."""
    from zolt_amd import api
    a, b = _rand(ob, 2000, 1 << 10), _rand(ob, 2001, 1 << 10)
    s = _rand(ob, 2002, 1)[0]
    pa, pb = api.DensePolynomial(a), api.DensePolynomial(b)
    assert np.array_equal(pa.add(pb).evaluations, ob.f_add(ob.FR, a, b))
    assert np.array_equal(pa.scale(s).evaluations, ob.fr_poly_scale(a, s))
    assert np.array_equal(pa.scale(s).evaluations, ob.f_mul(ob.FR, a, np.repeat(s[None, :], 1 << 10, axis=0)))
    for v in (0, 1, 5, 20):
        r, x = _rand(ob, 2010 + v, v), _rand(ob, 2020 + v, v)
        assert np.array_equal(api.EqPolynomial.mle(r, x), ob.fr_eq_mle(r, x))
        assert np.array_equal(api.EqPolynomial(r).evaluate(x), ob.fr_eq_mle(r, x))
    r = _rand(ob, 2030, 6)
    tab = zl.fr_eq_table(r)
    for idx in (0, 1, 37, 63):
        bits = U.fr([(idx >> (5 - j)) & 1 for j in range(6)])
        assert np.array_equal(api.EqPolynomial.mle(r, bits), tab[idx])
