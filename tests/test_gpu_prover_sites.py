"""SURVEY 8(f)3 — the prover fold sites with a REAL host transcript: Stage-1 [p0, p1, 2p1 - p0] rounds driven by the Keccak
transcript (src/zkvm/prover.zig:397-432, src/zkvm/r1cs/jolt_r1cs.zig:413-486), the RAF cubic round polynomial
(src/zkvm/ram/raf_checking.zig:335-445) and the Lasso address-round gather sums (src/zkvm/lasso/prover.zig:283-306), device
path against the oracle's restatement, bit for bit."""
import numpy as np
import pytest

from tests import util as U

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from oracle import binding as ob
    from zolt_amd import api, lib
    lib.init()
    return api, lib, ob


def _rand(ob, seed, n):
    return ob.f_to_mont(ob.FR, U.random_raw256(seed, n))


@pytest.mark.parametrize("v,rounds", [(0, 2), (1, 1), (2, 4), (5, 5), (13, 13), (16, 16)])
def test_stage1_rounds_with_keccak_transcript(env, v, rounds):
    """v = 13 is the captured run's log_constraints (logs/zolt.log:43). The transcript is seeded the way proveStage1's caller
    leaves it: domain "Jolt", some absorbed bytes, v "spartan_tau" challenges squeezed first."""
    api, lib, ob = env
    poly = _rand(ob, 3000 + v, 1 << v)
    ta, tb = api.Transcript(b"Jolt"), ob.Transcript(b"Jolt")
    ta.appendBytes(b"commitments" * 7); tb.append_bytes(b"commitments" * 7)
    for _ in range(v):
        assert np.array_equal(ta.challengeScalar(b"spartan_tau"), tb.challenge_scalar(b"spartan_tau"))
    got = api.proveStage1(poly, rounds, ta)
    wrp, wch, wfin = ob.stage1_prove(poly, rounds, tb)
    assert np.array_equal(got["round_polys"], wrp) and np.array_equal(got["challenges"], wch)
    assert np.array_equal(got["final_eval"], wfin)
    assert bytes(ta.state) == tb.state_bytes()[0]  # both transcripts end in the same state: later stages stay in sync


@pytest.mark.parametrize("log_k", [1, 4, 10, 16])
def test_raf_cubic_rounds(env, log_k):
    """Stage 2 as the reference runs it (log_k = 16 in the captured run, logs/zolt.log:41): every round's s(0..3), the Lagrange
    claim update, the bind, the final claim."""
    api, lib, ob = env
    ra = _rand(ob, 3100 + log_k, 1 << log_k)
    start = 0x7FFF8000
    # initial claim = sum_k ra(k) * unmap(k), unmap(k) = start + 8k (:321-330; UnmapPolynomial)
    claim = sum(U.fr_to_int(ra[k]) * (start + 8 * k) for k in range(1 << min(log_k, 10))) if log_k <= 10 else None
    if claim is None:
        claim = int(U.fr_to_int(_rand(ob, 1, 1)[0]))  # any claim: s(1) and s(3) are linear in it
    claim_l = api.fr_from_int(claim)
    prover = api.RafEvaluationProver(ra, start, log_k, claim_l)
    ta, tb = api.Transcript(b"Jolt"), ob.Transcript(b"Jolt")
    cur, bound, wclaim = ra.copy(), np.zeros((0, 4), dtype=np.uint64), claim_l
    for rd in range(log_k):
        got = prover.computeRoundPolynomialCubic()
        want = ob.raf_round_cubic(cur, start, bound, log_k, wclaim)
        assert np.array_equal(got, want), rd
        if rd == 0 and log_k <= 10:
            assert (U.fr_to_int(want[0]) + U.fr_to_int(want[1])) % api.R_MOD == claim % api.R_MOD
        ch = ta.challengeScalar(b"raf_round")
        assert np.array_equal(ch, tb.challenge_scalar(b"raf_round"))
        prover.updateClaim(got, ch)
        wclaim = ob.raf_update_claim(want, ch)
        assert np.array_equal(prover.current_claim, wclaim), rd
        prover.bindChallenge(ch)
        cur = ob.fr_bind_low(cur, ch)
        bound = np.concatenate([bound, ch[None, :]])
        if rd in (0, log_k // 2):  # RaPolynomial.finalClaim is evals[0] at any point, not only after the last bind (raf_checking.zig:179-185)
            assert np.array_equal(prover.getFinalClaim(), cur[0]), rd
    assert prover.isComplete() and np.array_equal(prover.getFinalClaim(), cur[0])
    if log_k <= 10:  # a true claim stays consistent: the final claim equals ra(r) * unmap(r)
        unmap_r = (start + 8 * sum(U.fr_to_int(b) << j for j, b in enumerate(bound))) % api.R_MOD
        assert U.fr_to_int(wclaim) == U.fr_to_int(cur[0]) * unmap_r % api.R_MOD
    prover.deinit()


@pytest.mark.parametrize("log_k", [0, 1, 7, 13, 16])
def test_raf_initial_claim_on_the_device(env, log_k):
    """RafEvaluationProver.computeInitialClaim (src/zkvm/ram/raf_checking.zig:312-321): sum_k ra(k) * F.fromU64(start + 8 k), one pass over
    the resident table, against exact integers; the prover built without a claim starts from it, and its first round polynomial satisfies
    s(0) + s(1) = claim with s(0) from the oracle."""
    api, lib, ob = env
    K = 1 << log_k
    ra = _rand(ob, 3150 + log_k, K)
    for start in (0x7FFF8000, 0, (1 << 64) - 8 * K):
        want = sum(U.fr_to_int(ra[k]) * (start + 8 * k) for k in range(K)) % api.R_MOD
        prover = api.RafEvaluationProver(ra, start, log_k)
        assert U.fr_to_int(prover.current_claim) == want, (log_k, start)
        assert np.array_equal(prover.computeInitialClaim(), api.fr_from_int(want))
        if log_k >= 1 and start < (1 << 63):
            got = prover.computeRoundPolynomialCubic()
            assert np.array_equal(got, ob.raf_round_cubic(ra, start, np.zeros((0, 4), dtype=np.uint64), log_k, api.fr_from_int(want)))
        prover.deinit()
    if K > 1:  # start + 8 (K - 1) = 2^64: the reference's u64 sum (UnmapPolynomial.evaluateAtIndex) would overflow
        s = lib.SumcheckSession.open(ra, lib.SC_LOW_PAIR)
        with pytest.raises(lib.ZgError):
            s.raf_claim((1 << 64) - 8 * K + 8, 8)
        s.close()


def test_raf_round_rejects_overflow_and_wrong_layout(env):
    api, lib, ob = env
    s = lib.SumcheckSession.open(_rand(ob, 1, 16), lib.SC_HIGH_HALF)
    with pytest.raises(lib.ZgError):
        s.raf_round(api.fr_from_int(1), 8)
    s.close()
    s = lib.SumcheckSession.open(_rand(ob, 1, 16), lib.SC_LOW_PAIR)
    with pytest.raises(lib.ZgError):
        s.raf_round(api.fr_from_int(1), 1 << 62)
    s.close()


@pytest.mark.parametrize("n", [1, 255, 4096, 100000])
def test_lasso_address_round_sums(env, n):
    api, lib, ob = env
    eq = _rand(ob, 3200 + n % 97, n)
    idx = U.splitmix64(77 + n, 2 * n).reshape(n, 2)
    rounds = api.LassoAddressRounds(eq, idx)
    for bit in (0, 1, 17, 63, 64, 100, 127):
        w0, w1 = ob.lasso_address_sums(eq, idx, bit)
        c = rounds.computeAddressRoundPoly(bit)
        assert np.array_equal(c[0], w0) and np.array_equal(c[1], ob.f_sub(ob.FR, w1, w0)) and not c[2].any(), bit
        h0, h1 = lib.fr_bit_split_sums(eq, idx, bit)
        assert np.array_equal(h0, w0) and np.array_equal(h1, w1)
    # the two sums always add up to the claim (sum of all eq values): the invariant the reference prints as sumcheck_ok
    tot = ob.fr_sum_halves(np.concatenate([eq, np.zeros_like(eq)]))[0]
    w0, w1 = lib.fr_bit_split_sums(eq, idx, 5)
    assert np.array_equal(ob.f_add(ob.FR, w0, w1), tot)


def _lasso_case(ob, seed, log_T, log_K, n):
    w = _rand(ob, seed, log_T)
    idx = U.splitmix64(seed + 1, 2 * max(n, 1)).reshape(-1, 2)[:n].copy()
    if log_K < 128:  # indices below 2^log_K, as the reference's callers produce them
        mask = (1 << log_K) - 1
        idx[:, 0] &= np.uint64(mask & (2**64 - 1))
        idx[:, 1] &= np.uint64(mask >> 64)
    return w, idx


def test_lasso_prover_reference_inline_tests(env):
    """src/zkvm/lasso/prover.zig:553-688 restated against api.LassoProver (one device session for both phases)."""
    api, lib, ob = env
    F, P = api.fr_from_int, api.R_MOD
    idx = np.array([[0, 0], [1, 0], [2, 0], [3, 0]], dtype=np.uint64)
    p = api.LassoProver(idx, 2, 3, np.stack([F(2), F(3)]))
    assert p.round == 0 and p.isAddressPhase() and not p.isComplete()
    assert len(p.computeRoundPolynomial()) > 0
    p.receiveChallenge(F(7))
    assert p.round == 1
    p.deinit()
    p = api.LassoProver(idx, 2, 3, np.stack([F(2), F(3)]))
    for rnd in range(5):
        claim = api.fr_to_int(p.current_claim)
        c0, c1, c2 = (api.fr_to_int(c) for c in p.computeRoundPolynomial())
        assert (c0 + c0 + c1 + c2) % P == claim
        ch = rnd + 10
        p.receiveChallenge(F(ch))
        assert api.fr_to_int(p.current_claim) == (c0 + c1 * ch + c2 * ch * ch) % P
    assert p.isComplete()
    p.deinit()


@pytest.mark.parametrize("log_T,log_K,n", [(0, 3, 1), (1, 1, 2), (2, 3, 4), (5, 16, 21), (8, 64, 256), (10, 128, 1000), (14, 32, 16384), (16, 16, 40000)])
def test_lasso_prover_vs_oracle(env, log_T, log_K, n):
    """every round polynomial, claim and the live eq_evals array of both phases against the oracle's restatement; random challenges"""
    api, lib, ob = env
    w, idx = _lasso_case(ob, 5100 + log_T, log_T, log_K, n)
    g, o = api.LassoProver(idx, log_T, log_K, w), ob.LassoProver(idx, log_T, log_K, w)
    assert np.array_equal(g.eq_evals(), o.eq_evals) and np.array_equal(g.computeInitialClaim(), o.current_claim)
    chal = _rand(ob, 5200 + log_T, log_T + log_K)
    for rnd in range(log_T + log_K):
        assert g.isAddressPhase() == o.isAddressPhase()
        assert np.array_equal(g.computeRoundPolynomial(), o.computeRoundPolynomial()), rnd
        g.receiveChallenge(chal[rnd])
        o.receiveChallenge(chal[rnd])
        assert np.array_equal(g.current_claim, o.current_claim), rnd
        if rnd % 7 == 0 or rnd >= log_K:
            assert np.array_equal(g.eq_evals(), o.eq_evals[:o.eq_evals_len]), rnd
    assert g.isComplete() and o.isComplete()
    assert np.array_equal(g.computeRoundPolynomial(), o.computeRoundPolynomial())  # the num_cycles <= 1 branch (:325-333)
    assert np.array_equal(g.getFinalEval(), o.getFinalEval())
    g.deinit()


@pytest.mark.parametrize("log_T,log_K,n", [(2, 3, 4), (6, 16, 50), (12, 24, 4096)])
def test_run_lasso_prover_vs_oracle(env, log_T, log_K, n):
    """runLassoProver (:495-530) with the reference's own 64-bit challenge mixer (:533-551)"""
    api, lib, ob = env
    w, idx = _lasso_case(ob, 5300 + log_T, log_T, log_K, n)
    got, want = api.runLassoProver(idx, log_T, log_K, w), ob.run_lasso_prover(idx, log_T, log_K, w)
    for k in ("round_polys", "final_eval", "challenges"):
        assert np.array_equal(got[k], want[k]), k


def test_lasso_session_bit_ops_edge_cases(env):
    """zg_sumcheck_bit_round / bit_bind directly: non-zero padding past the lookups stays untouched and is part of the claim; a
    bit_round on a different bit than the one bit_bind prepared recomputes; invalid arguments are refused; the ordinary
    HIGH_HALF rounds continue on the same session afterwards."""
    api, lib, ob = env
    n, n_idx = 1 << 10, 700
    tab = _rand(ob, 5400, n)
    idx = U.splitmix64(5401, 2 * n_idx).reshape(n_idx, 2)
    d_idx = lib.DeviceBuffer.from_host(idx)
    s = lib.SumcheckSession.open(tab, lib.SC_HIGH_HALF)
    cur = tab.copy()
    for bit in (3, 4, 64, 127, 9):
        w0, w1 = ob.lasso_address_sums(cur[:n_idx], idx, bit)
        g0, g1 = s.bit_round(d_idx.ptr, n_idx, bit)
        assert np.array_equal(g0, w0) and np.array_equal(g1, w1), bit
        r = _rand(ob, 5410 + bit, 1)[0]
        claim = np.zeros(4, dtype=np.uint64)
        ob.lib.zo_lasso_receive_address(ob._p(cur), n, ob._p(idx), n_idx, bit, ob._p(r), ob._p(claim))
        assert np.array_equal(s.bit_bind(d_idx.ptr, n_idx, bit, r), claim), bit
        assert np.array_equal(s.read(), cur)
    g0, g1 = s.round_sums()
    w0, w1 = ob.fr_sum_halves(cur)
    assert np.array_equal(g0, w0) and np.array_equal(g1, w1)
    r = _rand(ob, 5420, 1)[0]
    s.bind(r)
    assert np.array_equal(s.read(), ob.fr_bind_high(cur, r))
    with pytest.raises(RuntimeError):
        s.bit_round(d_idx.ptr, n, 0)  # n_idx beyond the (folded) table
    with pytest.raises(RuntimeError):
        s.bit_round(d_idx.ptr, 10, 128)
    with pytest.raises(RuntimeError):
        s.bit_bind(0, 10, 1, r)
    s.close()
    d_idx.free()


@pytest.mark.parametrize("v", [0, 1, 2, 6, 13, 16])
def test_jolt_outer_prover_rounds(env, v):
    """JoltOuterProver's loop (src/zkvm/spartan/jolt_outer_prover.zig:148-262): [p(0), p(2)], the cubic form by linear extrapolation,
    the fold and the claim (= sum of the folded table), against the oracle's LowToHigh sums / two-product fold"""
    api, lib, ob = env
    n = 1 << v
    w = _rand(ob, 5600 + v, n)
    g = api.JoltOuterProver(w)
    cur = w.copy()
    tot = lambda t: ob.f_add(ob.FR, *[x[None, :] for x in ob.fr_sum_even_odd(t)])[0] if len(t) >= 2 else t[0]
    assert np.array_equal(g.current_claim, tot(cur))
    chal = _rand(ob, 5610 + v, v + 1)
    two, three = ob.f_from_u64(ob.FR, np.array([2], dtype=np.uint64)), ob.f_from_u64(ob.FR, np.array([3], dtype=np.uint64))
    for rnd in range(v + 1):
        if len(cur) >= 2:
            p0, p1 = ob.fr_sum_even_odd(cur)
            c1 = ob.f_sub(ob.FR, p1[None, :], p0[None, :])
            want2 = ob.f_sub(ob.FR, ob.f_add(ob.FR, p1[None, :], p1[None, :]), p0[None, :])[0]
            assert np.array_equal(g.computeRoundPoly(), np.stack([p0, want2]))
            cub = g.computeCubicRoundPoly()
            assert np.array_equal(cub[0], p0)
            assert np.array_equal(cub[1], ob.f_add(ob.FR, p0[None, :], ob.f_mul(ob.FR, c1, two))[0])
            assert np.array_equal(cub[2], ob.f_add(ob.FR, p0[None, :], ob.f_mul(ob.FR, c1, three))[0])
            cur = ob.fr_bind_low_2mul(cur, chal[rnd])  # (1-r)*lo + r*hi (:240-243)
        else:
            assert np.array_equal(g.computeRoundPoly()[0], g.current_claim) and not g.computeCubicRoundPoly()[1:].any()
        g.bindChallenge(chal[rnd])
        assert g.current_len == len(cur) and np.array_equal(g.current_claim, tot(cur))
    assert np.array_equal(g.getFinalEval(), cur[0])
    g.deinit()


def test_full_size_lasso_prover(env):
    """2^20 cycles (BASELINE config 3's table length), log_K = 16: every address and cycle round against the oracle, final evaluation"""
    api, lib, ob = env
    log_T, log_K, n = 20, 16, (1 << 20) - 12345
    w, idx = _lasso_case(ob, 9100, log_T, log_K, n)
    g, o = api.LassoProver(idx, log_T, log_K, w), ob.LassoProver(idx, log_T, log_K, w)
    assert np.array_equal(g.computeInitialClaim(), o.current_claim)
    chal = _rand(ob, 9101, log_T + log_K)
    for rnd in range(log_T + log_K):
        assert np.array_equal(g.computeRoundPolynomial(), o.computeRoundPolynomial()), rnd
        g.receiveChallenge(chal[rnd])
        o.receiveChallenge(chal[rnd])
        assert np.array_equal(g.current_claim, o.current_claim), rnd
    assert np.array_equal(g.getFinalEval(), o.getFinalEval())
    assert np.array_equal(g.eq_evals(), o.eq_evals[:1])
    g.deinit()


def test_lasso_rounds_of_the_captured_run_on_the_device(env, golden_dir):
    """The reference's captured LassoProver run (tests/golden/lasso_rounds.json, logs/zolt.log:430-970) against the DEVICE: all 24
    claim updates claim_after = c0 + c1 * challenge in one zg_field_op pass each (raw Montgomery limbs in, raw limbs out), and the
    shape of the run — 44 lookups in a 2^8-entry table: the cycle rounds 16 and 17 have an empty second half — reproduced by
    api.LassoProver (zero padding, HIGH_HALF folds) on arbitrary tables."""
    import json
    import os
    api, lib, ob = env
    d = json.load(open(os.path.join(golden_dir, "lasso_rounds.json")))

    def fe(h):
        return np.array([int(h[48:64], 16), int(h[32:48], 16), int(h[16:32], 16), int(h[0:16], 16)], dtype=np.uint64)
    R = d["rounds"]
    c0, c1, ch = (np.stack([fe(r[k]) for r in R]) for k in ("c0", "c1", "challenge"))
    after = lib.field_op(lib.FR, lib.OP_ADD, c0, lib.field_op(lib.FR, lib.OP_MUL, c1, ch))
    assert np.array_equal(after, np.stack([fe(r["claim_after"]) for r in R]))
    assert np.array_equal(after[:-1], np.stack([fe(r["claim"]) for r in R[1:]]))
    assert np.array_equal(lib.field_op(lib.FR, lib.OP_ADD, c0, lib.field_op(lib.FR, lib.OP_ADD, c0, c1)), np.stack([fe(r["claim"]) for r in R]))
    rng = np.random.default_rng(44)
    idx = np.zeros((44, 2), dtype=np.uint64)
    idx[:, 0] = rng.integers(0, 1 << 16, size=44, dtype=np.uint64)
    p = api.LassoProver(idx, d["log_T"], d["log_K"], _rand(ob, 4401, d["log_T"]))
    zero = np.zeros(4, dtype=np.uint64)
    empty_second = []
    for i in range(d["total_rounds"]):
        co = p.computeRoundPolynomial()
        if i >= d["log_K"] and np.array_equal(lib.field_op(lib.FR, lib.OP_ADD, co[0][None], co[1][None])[0], zero):
            empty_second.append(i)
        p.receiveChallenge(_rand(ob, 4500 + i, 1)[0])
    p.deinit()
    assert empty_second == [i for i, r in enumerate(R) if i >= d["log_K"] and not fe(r["p1"]).any()] == [16, 17]


@pytest.mark.parametrize("n_steps", [1, 2, 200, 256, 1 << 13, (1 << 20) - 3])
def test_stages_5_and_6_with_keccak_transcript(env, n_steps):
    """MultiStageProver.proveStage5 / proveStage6 (src/zkvm/prover.zig:829-1112): the register-eq table and the booleanity table through a
    HIGH_HALF device session against the restatement — challenges, [p(0), p(2)] of every round, the claim chain, the final claims and the
    transcripts' end state; 256 steps is the captured run's length (logs/zolt.log:1064)."""
    api, lib, ob = env
    rng = np.random.default_rng(n_steps)
    instr = rng.integers(0, 1 << 32, size=n_steps, dtype=np.uint64).astype(np.uint32)
    log_t = max((n_steps - 1).bit_length(), 1)
    ta, tb = api.Transcript(b"Jolt"), ob.Transcript(b"Jolt")
    ta.appendBytes(b"stages 1-4"); tb.append_bytes(b"stages 1-4")
    g5, w5 = api.proveStage5(instr, log_t, ta), ob.stage5_prove(instr, log_t, tb)
    g6, w6 = api.proveStage6(n_steps, ta), ob.stage6_prove(n_steps, tb)
    for g, w in ((g5, w5), (g6, w6)):
        assert g.keys() == w.keys()
        for k in w:
            assert np.array_equal(np.asarray(g[k]), np.asarray(w[k])), k
    assert len(g5["round_polys"]) == (0 if n_steps <= 1 else (n_steps - 1).bit_length())
    if n_steps > 1:  # the sumcheck relations the verifier uses: p(1) = (p(0) + p(2)) / 2, p(0) + p(1) = claim
        claim = ob.fr_to_int(g5["initial_claim"])
        for rp, nxt in zip(g5["round_polys"], g5["claims"]):
            p0, p2 = ob.fr_to_int(rp[0]), ob.fr_to_int(rp[1])
            assert (p0 + (p0 + p2) * pow(2, -1, ob._R_P)) % ob._R_P == claim
            claim = ob.fr_to_int(nxt)
        assert claim == ob.fr_to_int(g5["final_claim"])
    assert not g6["final_claim"].any() and not g6["round_polys"].any()
    assert bytes(ta.state) == tb.state_bytes()[0]


def test_stage5_empty_trace(env):
    api, lib, ob = env
    ta, tb = api.Transcript(b"Jolt"), ob.Transcript(b"Jolt")
    g, w = api.proveStage5(np.zeros(0, dtype=np.uint32), 3, ta), ob.stage5_prove(np.zeros(0, dtype=np.uint32), 3, tb)
    assert g["initial_claim"] is None and w["initial_claim"] is None and np.array_equal(g["r_register"], w["r_register"])
    assert api.proveStage6(0, ta)["initial_claim"] is None and ob.stage6_prove(0, tb)["initial_claim"] is None
    assert bytes(ta.state) == tb.state_bytes()[0]
