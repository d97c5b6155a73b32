"""Product-form sumcheck sessions (zg_psc_*) and the prover loops that sit on them — ValEvaluationProver
(src/zkvm/ram/val_evaluation.zig:545-700), ValFinalProver (ram/val_final.zig:144-230), OutputSumcheckProver (ram/output_check.zig:375-499),
InstructionLookupsClaimReductionProver (claim_reductions/instruction_lookups.zig:146-284), ProductVirtualRemainderProver
(spartan/product_remainder.zig:269-394) — device path against the oracle's restatement of each loop, bit for bit."""
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


def _rand(ob, seed, n, sparse=False):
    a = ob.f_to_mont(ob.FR, U.random_raw256(seed, n))
    if sparse and n:
        a[U.splitmix64(seed + 7, n) % np.uint64(3) == 0] = 0
    return a


def _claim_of(api, tables_int):
    acc = 0
    for row in zip(*tables_int):
        p = 1
        for v in row:
            p = p * v % api.R_MOD
        acc = (acc + p) % api.R_MOD
    return acc


@pytest.mark.parametrize("p,q", [(1, 0), (2, 0), (3, 0), (4, 0), (0, 1), (0, 3), (1, 3), (2, 2), (3, 1), (4, 4)])
def test_round_evals_against_big_int_model(env, p, q):
    """zg_psc_round_evals at every (p, q) shape against plain Python integers, then a fold, then the evaluations again"""
    api, lib, ob = env
    P = api.R_MOD
    k, n = 8, 64
    tabs = [_rand(ob, 7000 + 10 * p + q + j, n, sparse=(j % 3 == 1)) for j in range(k)]
    ints = [[api.fr_to_int(x) for x in t] for t in tabs]
    prod_idx = [(3 * j + 1) % k for j in range(p)]
    lin_idx = [(5 * m + 2) % k for m in range(q)]
    coeff = _rand(ob, 7100 + p + q, max(q, 1))[:q]
    cint = [api.fr_to_int(c) for c in coeff]
    s = lib.ProductSumcheckSession.open(tabs)
    assert len(s) == n and s.tables() == k
    for rnd in range(3):
        half = len(s) // 2
        want = []
        for t in range(4):
            acc = 0
            for g in range(half):
                f = lambda T: (T[2 * g] + t * (T[2 * g + 1] - T[2 * g])) % P
                v = 1
                for j in prod_idx:
                    v = v * f(ints[j]) % P
                if q:
                    v = v * (sum(c * f(ints[m]) for c, m in zip(cint, lin_idx)) % P) % P
                acc = (acc + v) % P
            want.append(acc)
        got = s.round_evals(prod_idx, lin_idx, coeff if q else None)
        assert [api.fr_to_int(x) for x in got] == want, (p, q, rnd)
        r = _rand(ob, 7200 + rnd, 1)[0]
        ri = api.fr_to_int(r)
        s.bind(r)
        ints = [[(T[2 * i] + ri * (T[2 * i + 1] - T[2 * i])) % P for i in range(half)] for T in ints]
        for j in (0, k - 1):
            assert [api.fr_to_int(x) for x in s.read(j)] == ints[j]
    s.close()


def test_session_edges_and_errors(env):
    api, lib, ob = env
    t = _rand(ob, 7300, 2)
    s = lib.ProductSumcheckSession.open([t, t])
    with pytest.raises(RuntimeError):
        s.round_evals((0, 2))  # table index out of range
    with pytest.raises(RuntimeError):
        s.round_evals(())  # p + q == 0
    with pytest.raises(RuntimeError):
        s.final()  # two entries left
    s.bind(t[0])
    assert len(s) == 1
    f = s.final()
    assert np.array_equal(f[0], ob.fr_bind_low(t, t[0])[0]) and np.array_equal(f[0], f[1])
    with pytest.raises(RuntimeError):
        s.bind(t[0])
    with pytest.raises(RuntimeError):
        s.round_evals((0, 1))
    s.close()
    with pytest.raises(RuntimeError):
        lib.ProductSumcheckSession.open([_rand(ob, 7301, 3)])  # not a power of two
    with pytest.raises(RuntimeError):
        lib.ProductSumcheckSession.open([t] * 13)
    # device-pointer open: the same session contents
    d = [lib.DeviceBuffer.from_host(x) for x in (_rand(ob, 7302, 32), _rand(ob, 7303, 32))]
    s = lib.ProductSumcheckSession.open_dev([b.ptr for b in d], 32)
    assert np.array_equal(s.read(1), d[1].to_host().reshape(-1, 4))
    s.close()
    for b in d:
        b.free()


@pytest.mark.parametrize("v", [0, 1, 2, 5, 10, 14])
@pytest.mark.parametrize("three", [True, False])
def test_val_evaluation_and_val_final_provers(env, v, three):
    api, lib, ob = env
    n = 1 << v
    inc, wa = _rand(ob, 7400 + v, n, sparse=True), _rand(ob, 7410 + v, n)
    lt = _rand(ob, 7420 + v, n) if three else None
    claim = _rand(ob, 7430 + v, 1)[0]
    g = api.ValEvaluationProver(inc, wa, lt, claim) if three else api.ValFinalProver(inc, wa, claim)
    o = ob.ValEvaluationProver(inc, wa, lt, claim)
    ch = _rand(ob, 7440 + v, v + 1)
    for rnd in range(v + 1):  # one call past the last round: the single-entry branch (:559-565) and the no-op bind (:611-614)
        rp, wrp = g.computeRoundPolynomial(), o.computeRoundPolynomial()
        assert np.array_equal(rp, wrp), rnd
        g.bindChallengeWithPoly(ch[rnd], rp)
        o.bindChallengeWithPoly(ch[rnd], wrp)
        assert np.array_equal(g.current_claim, o.current_claim) and g.effectiveLen() == o.n
    for a, b in zip(g.getFinalClaims(), o.getFinalClaims()):
        assert np.array_equal(a, b)
    g.deinit()


@pytest.mark.parametrize("v", [1, 2, 6, 12, 16])
def test_output_sumcheck_prover(env, v):
    api, lib, ob = env
    n = 1 << v
    tabs = [_rand(ob, 7500 + 10 * j + v, n, sparse=(j == 1)) for j in range(5)]
    claim = _rand(ob, 7560 + v, 1)[0]
    g, o = api.OutputSumcheckProver(*tabs, claim), ob.OutputSumcheckProver(*tabs, claim)
    ch = _rand(ob, 7570 + v, v)
    for rnd in range(v):
        ev, wev = g.roundEvals(), o.roundEvals()
        assert np.array_equal(ev, wev), rnd
        assert np.array_equal(g.computeRoundPolynomial(), o.computeRoundPolynomial())
        g.bindChallenge(ch[rnd]); o.bindChallenge(ch[rnd])
        g.updateClaim(ev, ch[rnd]); o.updateClaim(wev, ch[rnd])
        assert np.array_equal(g.current_claim, o.current_claim) and g.current_size == o.current_size
    fg, fo = g.getFinalClaims(), o.getFinalClaims()
    assert all(np.array_equal(fg[k], fo[k]) for k in fo)
    g.deinit()


def test_output_sumcheck_of_the_captured_run(env, golden_dir):
    """The reference's own OutputSumcheck instance (Stage-2 instance 3 of its captured fibonacci run, logs/zolt.log "[ZOLT OUTPUT_CHECK]" /
    "OutputSumcheck:" lines of src/zkvm/ram/output_check.zig:100-365): every input is in the log or the ELF — r_address, the region
    bounds, the 13 program words — so the five 2^16-entry tables are rebuilt, the eq table comes from zg_fr_eq_table, the prover
    runs its 16 rounds on the device with the challenges the reference drew (batch rounds 8..23), and the five folded finals plus
    the running claim are held against what the reference printed. No oracle in between."""
    import json
    import os
    api, lib, ob = env
    d = json.load(open(os.path.join(golden_dir, "stage2_batched_rounds.json")))
    oc = d["output_check"]
    M = lambda h: api.fr_from_int(int.from_bytes(bytes.fromhex(h), "little"))
    elf = open(os.path.join(golden_dir, "fibonacci.elf"), "rb").read()
    tabs = U.output_check_tables_of_the_captured_run(oc, elf, api.fr_from_int, lib.fr_eq_table)
    assert int.from_bytes(bytes.fromhex(d["input_claims"][3]), "little") == 0
    g = api.OutputSumcheckProver(*tabs, api.fr_from_int(0))
    for k in range(d["num_rounds"][3]):
        ev = g.roundEvals()
        assert (api.fr_to_int(ev[0]) + api.fr_to_int(ev[1])) % api.R_MOD == api.fr_to_int(g.current_claim), k
        c = M(d["rounds"][oc["first_batch_round"] + k]["challenge"])
        g.updateClaim(ev, c)
        g.bindChallenge(c)
    f = g.getFinalClaims()
    assert sorted(f) == sorted(oc["final"]) and all(np.array_equal(f[k], M(h)) for k, h in oc["final"].items())
    assert np.array_equal(g.current_claim, M(d["instance_final_claims"][3]))
    g.deinit()


@pytest.mark.parametrize("v", [1, 3, 8, 13, 16])
def test_instruction_lookups_claim_reduction_prover(env, v):
    """with a consistent initial claim the s(0) + s(1) = claim chain holds through every round, as in the reference's run"""
    api, lib, ob = env
    n = 1 << v
    tabs = [_rand(ob, 7600 + 10 * j + v, n) for j in range(4)]
    gamma = _rand(ob, 7650 + v, 1)[0]
    gi = api.fr_to_int(gamma)
    if v <= 8:
        ints = [[api.fr_to_int(x) for x in t] for t in tabs]
        comb = [(o + gi * l + gi * gi * r) % api.R_MOD for o, l, r in zip(ints[1], ints[2], ints[3])]
        claim = api.fr_from_int(_claim_of(api, [ints[0], comb]))
    else:
        claim = _rand(ob, 7660 + v, 1)[0]
    g = api.InstructionLookupsClaimReductionProver(*tabs, gamma, claim)
    o = ob.InstructionLookupsClaimReduction(*tabs, gamma, claim)
    ch = _rand(ob, 7670 + v, v)
    for rnd in range(v):
        ev, wev = g.computeRoundPolynomialCubic(), o.computeRoundPolynomialCubic()
        assert np.array_equal(ev, wev), rnd
        g.bindChallenge(ch[rnd]); o.bindChallenge(ch[rnd])
        g.updateClaim(ev, ch[rnd]); o.updateClaim(wev, ch[rnd])
        assert np.array_equal(g.current_claim, o.current_claim)
    fg, fo = g.getOpeningClaims(), o.getOpeningClaims()
    assert all(np.array_equal(fg[k], fo[k]) for k in fo)
    if v <= 8:  # final claim = eq(r) * combined(r): the protocol was sound end to end
        f = g._s.final()
        fin = api.fr_to_int(f[0]) * (api.fr_to_int(f[1]) + gi * api.fr_to_int(f[2]) + gi * gi * api.fr_to_int(f[3])) % api.R_MOD
        assert fin == api.fr_to_int(g.current_claim)
    g.deinit()


@pytest.mark.parametrize("v", [1, 2, 3, 7, 8, 13, 16])
def test_product_virtual_remainder_prover(env, v):
    """Gruen form: device (t0, t_inf) under device-resident split-eq prefix tables + host cubic, against the oracle's nested loops"""
    api, lib, ob = env
    n = 1 << v
    left, right = _rand(ob, 7700 + v, n, sparse=True), _rand(ob, 7710 + v, n)
    tau, kernel = _rand(ob, 7720 + v, v), _rand(ob, 7730 + v, 1)[0]
    if v <= 8:  # consistent claim: sum_x eq(tau, x) * kernel * left * right
        eq = ob.fr_eq_table(tau, kernel)
        claim = api.fr_from_int(_claim_of(api, [[api.fr_to_int(x) for x in t] for t in (left, right, eq)]))
    else:
        claim = _rand(ob, 7740 + v, 1)[0]
    g, o = api.ProductVirtualRemainderProver(left, right, tau, kernel, claim), ob.ProductRemainderProver(left, right, tau, kernel, claim)
    ch = _rand(ob, 7750 + v, v)
    for rnd in range(v):
        ev, wev = g.roundEvals(), o.roundEvals()
        assert np.array_equal(ev, wev), rnd
        assert np.array_equal(g.computeRoundPolynomial(), o.computeRoundPolynomial())
        if v <= 8:
            assert (api.fr_to_int(ev[0]) + api.fr_to_int(ev[1])) % api.R_MOD == api.fr_to_int(g.current_claim)
        g.bindChallenge(ch[rnd]); o.bindChallenge(ch[rnd])
        g.updateClaim(ev, ch[rnd]); o.updateClaim(wev, ch[rnd])
        assert np.array_equal(g.current_claim, o.current_claim)
    assert np.array_equal(g.getFinalClaim(), o.getFinalClaim())
    assert np.array_equal(g.computeRoundPolynomial(), o.computeRoundPolynomial())  # no groups left: [claim, 0, 0]
    if v <= 8:
        assert api.fr_to_int(g.getFinalClaim()) * api.fr_to_int(g.split_eq.current_scalar) % api.R_MOD == api.fr_to_int(g.current_claim)
    g.deinit()


def test_gruen_round_direct(env):
    """zg_psc_round_gruen with p = 1, 3 and an E_out shorter than the pairs (the reference's g < num_groups / x_out < |E_out| guards)"""
    api, lib, ob = env
    P = api.R_MOD
    n = 64
    tabs = [_rand(ob, 7800 + j, n) for j in range(3)]
    ints = [[api.fr_to_int(x) for x in t] for t in tabs]
    e_out, e_in = _rand(ob, 7810, 4), _rand(ob, 7811, 4)  # 16 weights for 32 pairs: pairs 16..31 are outside
    d_out, d_in = lib.DeviceBuffer.from_host(e_out), lib.DeviceBuffer.from_host(e_in)
    s = lib.ProductSumcheckSession.open(tabs)
    for idx in ((1,), (0, 1, 2)):
        t0, ti = s.round_gruen(idx, d_out.ptr, 4, d_in.ptr, 4)
        w0 = wi = 0
        for g in range(16):
            w = api.fr_to_int(e_out[g >> 2]) * api.fr_to_int(e_in[g & 3]) % P
            a = b = w
            for j in idx:
                a = a * ints[j][2 * g] % P
                b = b * (ints[j][2 * g + 1] - ints[j][2 * g]) % P
            w0, wi = (w0 + a) % P, (wi + b) % P
        assert api.fr_to_int(t0) == w0 and api.fr_to_int(ti) == wi
    with pytest.raises(RuntimeError):
        s.round_gruen((0,), d_out.ptr, 4, d_in.ptr, 3)  # |E_in| not a power of two
    s.close()
    d_out.free(); d_in.free()


def test_fused_bind_keeps_every_table_and_every_spec_right(env):
    """zg_psc_bind folds AND prepares the next round's evaluations for the spec of the last round_evals call: changing the spec
    between rounds, Gruen rounds in between, binds without any evaluation before them, and tables outside the spec must all give
    the values of an unfused run (big-int model of all six tables)."""
    api, lib, ob = env
    P = api.R_MOD
    k, n = 6, 256
    tabs = [_rand(ob, 7900 + j, n, sparse=(j == 4)) for j in range(k)]
    ints = [[api.fr_to_int(x) for x in t] for t in tabs]
    coeff = _rand(ob, 7910, 2)
    cint = [api.fr_to_int(c) for c in coeff]
    e_out, e_in = _rand(ob, 7911, 16), _rand(ob, 7912, 8)
    d_out, d_in = lib.DeviceBuffer.from_host(e_out), lib.DeviceBuffer.from_host(e_in)

    def model(prod, lin):
        half = len(ints[0]) // 2
        res = []
        for t in range(4):
            acc = 0
            for g in range(half):
                f = lambda T: (T[2 * g] + t * (T[2 * g + 1] - T[2 * g])) % P
                v = 1
                for j in prod:
                    v = v * f(ints[j]) % P
                if lin:
                    v = v * (sum(c * f(ints[m]) for c, m in zip(cint, lin)) % P) % P
                acc = (acc + v) % P
            res.append(acc)
        return res

    s = lib.ProductSumcheckSession.open(tabs)
    plan = [((0, 1, 2), ()), ((0, 1, 2), ()), ((3,), (4, 5)), ("gruen",), ((0, 1, 2), ()), None, ((0, 1), (2, 3)), ((0, 1), (2, 3))]
    for rnd, step in enumerate(plan):
        if step is None:
            pass  # a bind with no evaluation since the last one (the cached spec is still the fourth step's)
        elif step[0] == "gruen":
            half = len(s) // 2
            no, ni = 4, 8
            t0, ti = s.round_gruen((1, 5), d_out.ptr, no, d_in.ptr, ni)
            w0 = wi = 0
            for g in range(min(half, no * ni)):
                w = api.fr_to_int(e_out[g >> 3]) * api.fr_to_int(e_in[g & 7]) % P
                w0 = (w0 + w * ints[1][2 * g] * ints[5][2 * g]) % P
                wi = (wi + w * (ints[1][2 * g + 1] - ints[1][2 * g]) * (ints[5][2 * g + 1] - ints[5][2 * g])) % P
            assert api.fr_to_int(t0) == w0 and api.fr_to_int(ti) == wi
        else:
            prod, lin = step
            got = s.round_evals(prod, lin, coeff if lin else None)
            assert [api.fr_to_int(x) for x in got] == model(prod, lin), rnd
            if rnd % 2:  # asking twice gives the same answer (the second call finds the mailbox already filled)
                assert np.array_equal(s.round_evals(prod, lin, coeff if lin else None), got)
        r = _rand(ob, 7920 + rnd, 1)[0]
        ri = api.fr_to_int(r)
        s.bind(r)
        half = len(ints[0]) // 2
        ints = [[(T[2 * i] + ri * (T[2 * i + 1] - T[2 * i])) % P for i in range(half)] for T in ints]
        for j in range(k):
            assert [api.fr_to_int(x) for x in s.read(j)] == ints[j], (rnd, j)
    assert len(s) == 1
    assert [api.fr_to_int(x) for x in s.final()] == [T[0] for T in ints]
    s.close()
    d_out.free(); d_in.free()


def _stage2_like_batch(api, ob, side, seed):
    """five instances with different round counts, as Stage 2 batches them (src/zkvm/batched_sumcheck.zig:1-21): ProductVirtualRemainder
    (6 rounds), RafEvaluation (9), ValEvaluation (11), OutputSumcheck (9), InstructionLookupsClaimReduction (6).
    side = "gpu": device-backed mirrors, "oracle": the oracle's restatements. -> list of (num_rounds, claim, round_fn, bind_fn, finals_fn)"""
    F = lambda s, n, sp=False: _rand(ob, seed + s, n, sparse=sp)
    out = []
    claims = F(90, 5)
    left, right, tau, kern = F(1, 64, True), F(2, 64), F(3, 6), F(4, 1)[0]
    ra = F(5, 512)
    inc, wa, lt = F(6, 2048, True), F(7, 2048), F(8, 2048)
    oc = [F(10 + j, 512, j == 1) for j in range(5)]
    il = [F(20 + j, 64) for j in range(4)]
    gamma = F(30, 1)[0]
    if side == "gpu":
        pv = api.ProductVirtualRemainderProver(left, right, tau, kern, claims[0])
        raf = api.RafEvaluationProver(ra, 0x7FFF8000, 9, claims[1])
        ve = api.ValEvaluationProver(inc, wa, lt, claims[2])
        op = api.OutputSumcheckProver(*oc, claims[3])
        ip = api.InstructionLookupsClaimReductionProver(*il, gamma, claims[4])
        raf_round = raf.computeRoundPolynomialCubic
    else:
        pv = ob.ProductRemainderProver(left, right, tau, kern, claims[0])
        ve = ob.ValEvaluationProver(inc, wa, lt, claims[2])
        op = ob.OutputSumcheckProver(*oc, claims[3])
        ip = ob.InstructionLookupsClaimReduction(*il, gamma, claims[4])

        class Raf:  # the oracle's RAF loop: table, bound challenges, claim (raf_checking.zig:335-445)
            def __init__(self):
                self.ra, self.bound, self.current_claim = ra.copy(), np.zeros((0, 4), dtype=np.uint64), claims[1].copy()

            def computeRoundPolynomialCubic(self):
                return ob.raf_round_cubic(self.ra, 0x7FFF8000, self.bound, 9, self.current_claim)

            def updateClaim(self, ev, ch):
                self.current_claim = ob.raf_update_claim(ev, ch)

            def bindChallenge(self, ch):
                self.ra = ob.fr_bind_low(self.ra, ch)
                self.bound = np.concatenate([self.bound, np.asarray(ch, dtype=np.uint64)[None, :]])

        raf = Raf()
        raf_round = raf.computeRoundPolynomialCubic
    last = {}

    def wrap(key, prover, round_fn, update=True):
        def rnd(_round):
            last[key] = round_fn()
            return last[key]

        def bind(ch):
            if update:
                prover.updateClaim(last[key], ch)
            prover.bindChallenge(ch)
        return rnd, bind

    r, b = wrap("pv", pv, pv.roundEvals)
    out.append((6, claims[0], r, b, lambda: pv.getFinalClaim()))
    r, b = wrap("raf", raf, raf_round)
    out.append((9, claims[1], r, b, lambda: raf.current_claim))
    out.append((11, claims[2], lambda _r: last.__setitem__("ve", ve.computeRoundPolynomial()) or last["ve"],
                lambda ch: ve.bindChallengeWithPoly(ch, last["ve"]), lambda: np.stack(ve.getFinalClaims())))
    r, b = wrap("op", op, op.roundEvals)
    out.append((9, claims[3], r, b, lambda: np.stack([v for v in op.getFinalClaims().values()])))
    r, b = wrap("ip", ip, ip.computeRoundPolynomialCubic)
    out.append((6, claims[4], r, b, lambda: np.stack([v for v in ip.getOpeningClaims().values()])))
    return out


@pytest.mark.parametrize("mode", ["proof_converter", "batched_sumcheck_zig"])
def test_batched_sumcheck_stage2_shape(env, mode):
    """BatchedSumcheckProver + generateBatchedProof (src/zkvm/batched_sumcheck.zig:77-430) over five device-backed instances that
    start at different rounds, with the Blake2b transcript: every compressed round polynomial, challenge, claim and every instance's
    final values against the oracle's restatement of the driver over the oracle's restatement of each instance."""
    api, lib, ob = env
    gpu, ora = _stage2_like_batch(api, ob, "gpu", 8000), _stage2_like_batch(api, ob, "oracle", 8000)
    p = api.BatchedSumcheckProver(mode)
    for nr, claim, rnd, bind, _ in gpu:
        p.addInstance(api.SumcheckInstance(nr, 3, claim, rnd, bind))
    t = api.Blake2bTranscript(b"Jolt")
    p.setupBatching(t)

    class Inst:
        def __init__(self, nr, claim, rnd, bind):
            self.num_rounds, self.input_claim, self.computeRoundPoly, self.bindChallenge = nr, claim, rnd, bind

    o = ob.BatchedSumcheck([Inst(nr, claim, rnd, bind) for nr, claim, rnd, bind, _ in ora], p.batching_coeffs, mode)
    assert p.numRounds() == 11 and np.array_equal(p.current_claim, o.current_claim)
    proof = api.generateBatchedProof(p, t)
    # the oracle side replays the same transcript: absorb its own polynomials, the challenges must come out the same
    t2 = api.Blake2bTranscript(b"Jolt")
    for _, claim, _, _, _ in ora:
        t2.appendScalar(claim)
    for _ in ora:
        t2.challengeScalarFull()
    for k in range(11):
        comp = o.computeRoundPolynomial()
        assert np.array_equal(comp, proof["round_polys"][k]), k
        t2.appendMessage(b"UniPoly_begin")
        for c in comp:
            t2.appendScalar(c)
        t2.appendMessage(b"UniPoly_end")
        ch = t2.challengeScalar()
        assert np.array_equal(ch, proof["challenges"][k])
        full = ob.decompress_round_poly(comp, o.current_claim)
        o.updateClaim(full, ch)
        o.bindChallenge(ch)
    assert np.array_equal(o.current_claim, proof["final_claim"])
    for (_, _, _, _, fg), (_, _, _, _, fo) in zip(gpu, ora):
        assert np.array_equal(fg(), fo())


def test_folded_tables_handed_to_another_session_inside_hbm(env):
    """zg_psc_table_dev + zg_psc_open_dev (stream NULL: own stream, copies complete on return) — how Stage 3's phase transitions re-open
    their provers over the folded witness columns: after a few binds the tables of one session become (in another order, beside a
    fresh one) the tables of a second, which then folds on by itself; the first may be closed at once"""
    api, lib, ob = env
    n, k = 512, 4
    tabs = [_rand(ob, 8700 + j, n) for j in range(k)]
    a = lib.ProductSumcheckSession.open(tabs)
    cur = [t.copy() for t in tabs]
    a.round_evals((0, 1))  # with a spec cached, the binds below are the fused fold + next-sums launches
    for i in range(3):
        r = _rand(ob, 8720 + i, 1)[0]
        a.bind(r)
        cur = [ob.fr_bind_low(t, r) for t in cur]
    m = len(a)
    fresh = _rand(ob, 8730, m)
    d_fresh = lib.DeviceBuffer.from_host(fresh)
    b = lib.ProductSumcheckSession.open_dev([d_fresh.ptr, a.table_dev(3), a.table_dev(1), a.table_dev(0)], m)
    a.close()
    d_fresh.free()
    now = [fresh, cur[3], cur[1], cur[0]]
    for j in range(4):
        assert np.array_equal(b.read(j), now[j]), j
    while len(b) > 1:
        want = ob.ValEvaluationProver(now[0], now[1], now[2], now[0][0]).computeRoundPolynomial()
        assert np.array_equal(b.round_evals((0, 1, 2)), want), len(b)
        r = _rand(ob, 8740 + len(b), 1)[0]
        b.bind(r)
        now = [ob.fr_bind_low(t, r) for t in now]
    assert np.array_equal(b.final(), np.stack([t[0] for t in now]))
    b.close()


def test_pooled_sessions_serve_smaller_shapes(env):
    """a closed session is kept and reused for a later one with fewer tables / a shorter length (its buffers keep their stride): the
    reused session must behave like a fresh one through evaluations, folds and finals"""
    api, lib, ob = env
    for k, n, seed in ((5, 256, 1), (3, 128, 2), (2, 64, 3), (5, 256, 4), (1, 64, 5)):
        tabs = [_rand(ob, 8100 + 10 * seed + j, n) for j in range(k)]
        s = lib.ProductSumcheckSession.open(tabs)
        assert len(s) == n and s.tables() == k
        cur = [t.copy() for t in tabs]
        idx = tuple(range(min(k, 3)))
        while len(s) > 1:
            want = ob.ValEvaluationProver(cur[0], cur[1 % k], cur[2 % k] if len(idx) == 3 else None, cur[0][0]).computeRoundPolynomial() if len(idx) >= 2 else None
            got = s.round_evals(idx)
            if want is not None:
                assert np.array_equal(got, want), (k, n, len(s))
            r = _rand(ob, 8150 + len(s), 1)[0]
            s.bind(r)
            cur = [ob.fr_bind_low(t, r) for t in cur]
            assert np.array_equal(s.read(k - 1), cur[k - 1])
        assert np.array_equal(s.final(), np.stack([t[0] for t in cur]))
        s.close()


@pytest.mark.parametrize("shape", [[(2, 0)], [(2, 0), (2, 0), (2, 0), (2, 0)], [(1, 4), (0, 1), (1, 1)], [(2, 2), (2, 2), (2, 2), (2, 2)], [(4, 4), (3, 0), (0, 3)]])
def test_round_expr_against_big_int_model(env, shape):
    """zg_psc_round_expr: sums of product terms of every (p, q) shape the Stage-3 provers use, against plain Python integers, across
    three rounds (the second and third come out of the fused bind)"""
    api, lib, ob = env
    P = api.R_MOD
    k, n = 12, 64
    tabs = [_rand(ob, 8300 + j, n, sparse=(j % 4 == 1)) for j in range(k)]
    ints = [[api.fr_to_int(x) for x in t] for t in tabs]
    terms, model = [], []
    for ti, (p, q) in enumerate(shape):
        prod_idx = tuple((5 * ti + 3 * j + 1) % k for j in range(p))
        lin_idx = tuple((7 * ti + 2 * m) % k for m in range(q))
        coeff = _rand(ob, 8350 + ti, max(q, 1))[:q]
        terms.append((prod_idx, lin_idx, coeff if q else None))
        model.append((prod_idx, lin_idx, [api.fr_to_int(c) for c in coeff]))
    s = lib.ProductSumcheckSession.open(tabs)
    for rnd in range(3):
        half = len(s) // 2
        want = []
        for t in range(4):
            acc = 0
            for g in range(half):
                f = lambda T: (T[2 * g] + t * (T[2 * g + 1] - T[2 * g])) % P
                for prod_idx, lin_idx, cint in model:
                    v = 1
                    for j in prod_idx:
                        v = v * f(ints[j]) % P
                    if lin_idx:
                        v = v * (sum(c * f(ints[m]) for c, m in zip(cint, lin_idx)) % P) % P
                    acc = (acc + v) % P
            want.append(acc)
        assert [api.fr_to_int(x) for x in s.round_expr(terms)] == want, (shape, rnd)
        r = _rand(ob, 8360 + rnd, 1)[0]
        ri = api.fr_to_int(r)
        s.bind(r)
        ints = [[(T[2 * i] + ri * (T[2 * i + 1] - T[2 * i])) % P for i in range(half)] for T in ints]
        for j in (0, 5, k - 1):
            assert [api.fr_to_int(x) for x in s.read(j)] == ints[j]
    with pytest.raises(RuntimeError):
        s.round_expr([((0, 12), (), None)])
    with pytest.raises(RuntimeError):
        s.round_expr([])
    with pytest.raises(RuntimeError):
        s.round_expr([((), (), None)])
    s.close()


@pytest.mark.parametrize("v", [1, 2, 5, 10, 14])
def test_instruction_input_prover(env, v):
    """InstructionInputProver (src/zkvm/spartan/stage3_prover.zig:2029-2150): ten tables, four product terms in one pass per round"""
    api, lib, ob = env
    n = 1 << v
    tabs = [_rand(ob, 8400 + 10 * j + v, n, sparse=(j in (0, 2, 4, 6))) for j in range(10)]
    gamma = _rand(ob, 8490 + v, 1)[0]
    g = api.InstructionInputProver(tabs, gamma)
    cur = [t.copy() for t in tabs]
    claim = _rand(ob, 8495 + v, 1)[0]
    ch = _rand(ob, 8496 + v, v)
    for rnd in range(v):
        got, want = g.computeRoundEvals(claim), ob.instruction_input_round(cur, gamma, claim)
        assert np.array_equal(got, want), rnd
        g.bind(ch[rnd])
        cur = [ob.fr_bind_low(t, ch[rnd]) for t in cur]
        claim = ob.raf_update_claim(want, ch[rnd])
    fc = g.finalClaims()
    assert all(np.array_equal(fc[name], cur[j][0]) for j, name in enumerate(api.InstructionInputProver.NAMES))
    g.deinit()


@pytest.mark.parametrize("v", [1, 3, 8, 12])
def test_shift_sumcheck_rounds_both_phases(env, v):
    """ShiftSumcheckProver's phase-1 (four P*Q pairs) and phase-2 (eq_out * (...) + gamma^4 (1 - noop) eq_prod) loops
    (stage3_prover.zig:1343-1500,1782-1817) over given tables"""
    api, lib, ob = env
    n = 1 << v
    tabs = [_rand(ob, 8500 + 10 * j + v, n) for j in range(8)]
    p1 = api.ShiftSumcheckRounds(tabs)
    cur = [t.copy() for t in tabs]
    ch = _rand(ob, 8590 + v, v)
    for rnd in range(v):
        want = ob.shift_phase1_round(cur[0::2], cur[1::2], len(cur[0]))
        assert np.array_equal(p1.computeRoundEvals(None), want), rnd
        p1.bind(ch[rnd])
        cur = [ob.fr_bind_low(t, ch[rnd]) for t in cur]
    assert all(np.array_equal(a, b) for a, b in zip(p1.tables(), cur))
    p1.deinit()
    tabs = [_rand(ob, 8600 + 10 * j + v, n, sparse=(j == 6)) for j in range(7)]
    gp = _rand(ob, 8690 + v, 5)
    p2 = api.ShiftSumcheckRounds(tabs, phase2=True, gamma_powers=gp)
    cur = [t.copy() for t in tabs]
    claim = _rand(ob, 8695 + v, 1)[0]
    for rnd in range(v):
        want = ob.shift_phase2_round(cur, gp, claim)
        assert np.array_equal(p2.computeRoundEvals(claim), want), rnd
        p2.bind(ch[rnd])
        cur = [ob.fr_bind_low(t, ch[rnd]) for t in cur]
        claim = want[2]  # any value: the claim only enters p(1) = claim - p(0)
    p2.deinit()


@pytest.mark.parametrize("v", [1, 4, 9])
def test_registers_claim_reduction_rounds(env, v):
    """RegistersClaimReductionProver's two loops (stage3_prover.zig:2326-2481)"""
    api, lib, ob = env
    n = 1 << v
    gamma = _rand(ob, 8700 + v, 1)[0]
    claim = _rand(ob, 8701 + v, 1)[0]
    ch = _rand(ob, 8702 + v, v)
    for phase2, k in ((False, 2), (True, 4)):
        tabs = [_rand(ob, 8710 + 10 * j + v + (50 if phase2 else 0), n) for j in range(k)]
        g = api.RegistersClaimReductionRounds(tabs, gamma, phase2=phase2)
        cur = [t.copy() for t in tabs]
        for rnd in range(v):
            assert np.array_equal(g.computeRoundEvals(claim), ob.registers_cr_round(phase2, cur, gamma, claim)), (phase2, rnd)
            g.bind(ch[rnd])
            cur = [ob.fr_bind_low(t, ch[rnd]) for t in cur]
        if phase2:
            fc = g.finalClaims()
            assert np.array_equal(fc["rd_write_value"], cur[1][0]) and np.array_equal(fc["rs2_value"], cur[3][0])
        g.deinit()


def test_full_size_val_evaluation_and_instruction_input(env):
    """BASELINE config 3's size (2^20 entries) on the product-form path: every round of a three-table ValEvaluation and five rounds of the
    ten-table InstructionInput against the oracle (the per-thread lazy sums are flushed several times at this length), then the
    protocol's own identity on the final values"""
    api, lib, ob = env
    v = 20
    n = 1 << v
    inc, wa, lt = _rand(ob, 9000, n, sparse=True), _rand(ob, 9001, n), _rand(ob, 9002, n)
    claim = _rand(ob, 9003, 1)[0]
    g, o = api.ValEvaluationProver(inc, wa, lt, claim), ob.ValEvaluationProver(inc, wa, lt, claim)
    ch = _rand(ob, 9004, v)
    for rnd in range(v):
        rp, wrp = g.computeRoundPolynomial(), o.computeRoundPolynomial()
        assert np.array_equal(rp, wrp), rnd
        g.bindChallengeWithPoly(ch[rnd], rp)
        o.bindChallengeWithPoly(ch[rnd], wrp)
    assert all(np.array_equal(a, b) for a, b in zip(g.getFinalClaims(), o.getFinalClaims()))
    g.deinit()
    m = 1 << 18
    tabs = [_rand(ob, 9010 + j, m, sparse=(j % 2 == 0)) for j in range(10)]
    gamma = _rand(ob, 9030, 1)[0]
    p = api.InstructionInputProver(tabs, gamma)
    cur = tabs
    for rnd in range(5):
        want = ob.instruction_input_round(cur, gamma, claim)
        assert np.array_equal(p.computeRoundEvals(claim), want), rnd
        p.bind(ch[rnd])
        cur = [ob.fr_bind_low(t, ch[rnd]) for t in cur]
        claim = ob.raf_update_claim(want, ch[rnd])
    p.deinit()


@pytest.mark.parametrize("mask", [0b0101, 0b1101, 0b0111, 0b0001, 0b1000, 0b1010])
def test_set_points_skips_only_what_was_not_asked_for(env, mask):
    """zg_psc_set_points: the wanted evaluations are the ones an all-points session gives — for the plain product form, the
    product-times-combination form and the multi-term form, straight and through the evaluations fused into the bind — and the
    others come back as zero; changing the mask between two calls on the same tables recomputes."""
    api, lib, ob = env
    v = 9
    tabs = [_rand(ob, 9800 + j, 1 << v) for j in range(6)]
    co = _rand(ob, 9810, 3)
    terms = [((0, 1), (2, 3), co[:2]), ((4,), (), None), ((), (5, 1), co[1:3])]
    full, part = lib.ProductSumcheckSession.open(tabs), lib.ProductSumcheckSession.open(tabs)
    part.set_points(mask)
    zero = np.zeros(4, dtype=np.uint64)

    def check(got, want):
        for t in range(4):
            assert np.array_equal(got[t], want[t] if (mask >> t) & 1 else zero), t

    ch = _rand(ob, 9820, 4)
    for rnd in range(4):
        for call in (lambda s: s.round_evals((0, 1, 2)), lambda s: s.round_evals((3,), (0, 4, 5), co), lambda s: s.round_expr(terms)):
            check(call(part), call(full))
        # the last description is the one the bind fuses; ask for it again after the fold
        full.bind(ch[rnd]); part.bind(ch[rnd])
        check(part.round_expr(terms), full.round_expr(terms))
        for j in range(6):
            assert np.array_equal(part.read(j), full.read(j))
    # same description, other mask: not served from the mailbox of the previous call
    want = full.round_expr(terms)
    part.set_points(0xF)
    assert np.array_equal(part.round_expr(terms), want)
    part.set_points(0b0010)
    got = part.round_expr(terms)
    assert np.array_equal(got[1], want[1]) and not got[0].any() and not got[2].any() and not got[3].any()
    for bad in (0, 16, 255):
        with pytest.raises(RuntimeError):
            part.set_points(bad)
    full.close(); part.close()
    # a pooled handle starts at all four points again
    again = lib.ProductSumcheckSession.open(tabs)
    assert np.array_equal(again.round_expr(terms), lib.ProductSumcheckSession.open(tabs).round_expr(terms))
    assert again.round_expr(terms)[3].any()
    again.close()


def test_round_expr_terms_that_share_a_linear_combination(env):
    """psc_expr_kernel reuses a term's linear combination (and, in the fused bind, the fold of its tables) when the NEXT term names the
    same tables with the same coefficients: shared between neighbours, between all four, broken by a different coefficient or a
    different table, and a term without a combination in between — against the big-int model over three rounds (2 fused)."""
    api, lib, ob = env
    P = api.R_MOD
    k, n = 8, 128
    tabs = [_rand(ob, 10100 + j, n) for j in range(k)]
    co = _rand(ob, 10110, 4)
    one = api.fr_from_int(1)
    cases = [
        [((0, 1), (6, 7), co[:2]), ((2, 3), (6, 7), co[:2]), ((0, 2), (6, 7), co[2:4]), ((1, 3), (6, 7), co[2:4])],      # InstructionInput's pattern
        [((0,), (6, 7), co[:2]), ((1,), (6, 7), co[:2]), ((2,), (6, 7), co[:2]), ((3,), (6, 7), co[:2])],               # all four share
        [((0,), (6, 7), co[:2]), ((1,), (7, 6), co[:2]), ((2,), (6, 7), np.stack([co[0], co[2]])), ((3,), (6,), co[:1])],  # never the same
        [((0,), (5, 6), np.stack([one, co[1]])), ((1, 2), (), None), ((3,), (5, 6), np.stack([one, co[1]])), ((4,), (5, 6), np.stack([one, co[1]]))],
        [((), (6, 7), co[:2]), ((), (6, 7), co[:2])],                                                                    # the combination alone, twice
    ]
    for ci, terms in enumerate(cases):
        ints = [[api.fr_to_int(x) for x in t] for t in tabs]
        s = lib.ProductSumcheckSession.open(tabs)
        for rnd in range(3):
            half = len(s) // 2
            want = []
            for t in range(4):
                acc = 0
                for g in range(half):
                    f = lambda T: (T[2 * g] + t * (T[2 * g + 1] - T[2 * g])) % P
                    for prod_idx, lin_idx, coeff in terms:
                        v = 1
                        for j in prod_idx:
                            v = v * f(ints[j]) % P
                        if lin_idx:
                            v = v * (sum(api.fr_to_int(c) * f(ints[m]) for c, m in zip(coeff, lin_idx)) % P) % P
                        acc = (acc + v) % P
                want.append(acc)
            assert [api.fr_to_int(x) for x in s.round_expr(terms)] == want, (ci, rnd)
            r = _rand(ob, 10120 + rnd, 1)[0]
            if rnd == 1:
                r[:2] = 0
                r[3] &= np.uint64((1 << 61) - 1)
            ri = api.fr_to_int(r)
            s.bind(r)
            ints = [[(T[2 * i] + ri * (T[2 * i + 1] - T[2 * i])) % P for i in range(half)] for T in ints]
            for j in range(k):
                assert [api.fr_to_int(x) for x in s.read(j)] == ints[j], (ci, rnd, j)
        s.close()


@pytest.mark.parametrize("mask", [0xF, 0b1101, 0b0101])
def test_round_expr_pair_sum_terms(env, mask):
    """ZG_PSC_PAIR_SUM: (T[p0] T[p1] + T[p2] T[p3]) * L as ONE term — alone, next to ordinary terms, two of them sharing the combination
    (InstructionInput's form) — against the big-int model over three rounds (two out of the fused bind), under a point mask."""
    api, lib, ob = env
    P = api.R_MOD
    k, n = 10, 128
    tabs = [_rand(ob, 10200 + j, n, sparse=(j in (0, 4))) for j in range(k)]
    co = _rand(ob, 10210, 4)
    one = api.fr_from_int(1)
    cases = [
        [((4, 5, 6, 7), (8, 9), np.stack([one, co[0]]), True), ((0, 1, 2, 3), (8, 9), np.stack([co[1], co[2]]), True)],
        [((0, 1, 2, 3), (8,), co[:1], True)],
        [((0, 1), (), None), ((2, 3, 4, 5), (6, 7, 8), co[:3], True), ((9,), (6, 7, 8), co[:3])],
        [((0, 1, 0, 1), (2, 3), co[:2], True), ((0, 1, 2, 3), (2, 3), co[:2], True)],  # tables named more than once
        [((0, 1, 2, 3), (), None, True), ((4, 5, 6, 7), (), None, True)],               # no weight: ShiftSumcheck phase 1
        [((0, 1, 2, 3), (), None, True)],
    ]
    zero = [0, 0, 0, 0]
    for ci, terms in enumerate(cases):
        ints = [[api.fr_to_int(x) for x in t] for t in tabs]
        s = lib.ProductSumcheckSession.open(tabs)
        s.set_points(mask)
        for rnd in range(3):
            half = len(s) // 2
            want = []
            for t in range(4):
                acc = 0
                for g in range(half):
                    f = lambda T: (T[2 * g] + t * (T[2 * g + 1] - T[2 * g])) % P
                    for term in terms:
                        prod_idx, lin_idx, coeff = term[:3]
                        if len(term) > 3:
                            v = (f(ints[prod_idx[0]]) * f(ints[prod_idx[1]]) + f(ints[prod_idx[2]]) * f(ints[prod_idx[3]])) % P
                        else:
                            v = 1
                            for j in prod_idx:
                                v = v * f(ints[j]) % P
                        if lin_idx:
                            v = v * (sum(api.fr_to_int(c) * f(ints[m]) for c, m in zip(coeff, lin_idx)) % P) % P
                        acc = (acc + v) % P
                want.append(acc if (mask >> t) & 1 else 0)
            assert [api.fr_to_int(x) for x in s.round_expr(terms)] == want, (ci, rnd)
            r = _rand(ob, 10220 + rnd, 1)[0]
            if rnd == 0:
                r[:2] = 0
                r[3] &= np.uint64((1 << 61) - 1)
            ri = api.fr_to_int(r)
            s.bind(r)
            ints = [[(T[2 * i] + ri * (T[2 * i + 1] - T[2 * i])) % P for i in range(half)] for T in ints]
            for j in range(k):
                assert [api.fr_to_int(x) for x in s.read(j)] == ints[j], (ci, rnd, j)
        s.close()
    s = lib.ProductSumcheckSession.open(tabs)
    for bad in ([((0, 1, 2), (8,), co[:1], True)], [((0, 1), (), None, True)], [((0, 1), (8,), co[:1], True)]):
        with pytest.raises(RuntimeError):
            s.round_expr(bad)  # a pair sum needs exactly four tables
    s.close()


def test_stage2_product_virtualisation_of_the_captured_run_on_the_device(golden_dir):
    """the product-virtualisation instance of the reference's captured Stage 2 on the DEVICE, from the witnesses regenerated out of the ELF:
    t1 at the four targets (a product-sum launch over two-cycle windows), the fused tables built on the device (sliding-window affine maps)
    into the prover's session, eight Gruen rounds — the printed extended evaluations, 13 coefficients, input and final claim, full width"""
    from zolt_amd import api, lib
    from tests.test_transcript_host import check_stage2_product_virtual_against_the_captured_run
    lib.init(0)
    p = check_stage2_product_virtual_against_the_captured_run(api.productVirtualExtendedEvals, api.buildUniskipFirstRoundPoly,
                                                              api.productVirtualRemainderProverFromWitnesses, golden_dir, api.fr_from_int, api.fr_to_int)
    p.deinit()


@pytest.mark.parametrize("n", [1, 2, 77, 1024])
def test_product_virtual_from_witnesses_against_the_restatement(n):
    """random witnesses (not a satisfying assignment): extended evaluations and the fused tables / rounds, device against restatement"""
    from oracle import binding as ob
    from zolt_amd import api, lib
    from tests.test_transcript_host import random_cycle_witnesses
    lib.init(0)
    w = random_cycle_witnesses(900 + n, n)
    nv = max(n - 1, 0).bit_length()
    r = ob.f_to_mont(ob.FR, U.random_raw256(990 + n, 2 * nv + 4))
    tau, r0, claim, chals = r[:nv + 1], r[nv + 1], r[nv + 2], r[nv + 3:]
    assert np.array_equal(api.productVirtualExtendedEvals(w, tau), ob.product_virtual_extended_evals(w, tau))
    g, o = api.productVirtualRemainderProverFromWitnesses(w, r0, tau, claim), ob.product_remainder_prover_from_witness(w, r0, tau, claim)
    assert np.array_equal(g._s.read(0), o.left) and np.array_equal(g._s.read(1), o.right)
    for k in range(nv):
        eg, eo = g.roundEvals(), o.roundEvals()
        assert np.array_equal(eg, eo), k
        for p in (g, o):
            p.updateClaim(eo, chals[k])
            p.bindChallenge(chals[k])
    assert np.array_equal(g.getFinalClaim(), o.getFinalClaim())
    g.deinit()


def test_stage2_batched_proof_of_the_captured_run_on_the_device(golden_dir):
    """The reference's captured Stage-2 batched sumcheck with all five instances on the DEVICE, built from inputs (witnesses and memory
    access regenerated from the ELF): product virtualisation (fused tables by sliding-window affine maps), RAF (cubic round kernel),
    RAM read/write checking (the rwc session), output check and instruction-lookups claim reduction (product sessions), combined by
    api.BatchedSumcheckProver under the logged coefficients — all 24 compressed round polynomials, the claims between them and the output
    claim, full width."""
    from zolt_amd import api, lib
    from tests.test_transcript_host import check_stage2_batch_of_the_captured_run_from_inputs
    lib.init(0)

    def driver(insts, coeffs):
        p = api.BatchedSumcheckProver("proof_converter")
        for nr, claim, rnd, bind, _ in insts:
            p.addInstance(api.SumcheckInstance(nr, 3, claim, rnd, bind))
        p.batching_coeffs = list(coeffs)
        p.current_claim = p.batchedClaim()
        return p
    check_stage2_batch_of_the_captured_run_from_inputs("gpu", golden_dir, driver, api.decompressRoundPoly, api.fr_to_int)
