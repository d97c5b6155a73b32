"""SURVEY 8(f)3 — Stage 3 of the captured run on the device: the round loops of ShiftSumcheck (both phases), InstructionInput and
RegistersClaimReduction (both phases) run on product-form device sessions over the tables the reference builds
(src/zkvm/spartan/stage3_prover.zig), from the committed ELF and the logged challenges; every one of the eight combined round
polynomials, ShiftSumcheck's own p(0) / p(1), the claims and the final openings must be the printed bytes
(tests/golden/stage3_batched_rounds.json). The prefix / suffix tables and the phase-2 eq tables are small (sqrt(T) entries) host work —
built here by the oracle's restatement, as the device classes' docstrings say; the cycle-length tables (the witness columns, eq tables)
are folded on the device across the phase transition."""
import numpy as np
import pytest

from tests import test_transcript_host as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from oracle import binding as ob
    from zolt_amd import api, lib
    lib.init()
    return api, lib, ob


def test_stage3_of_the_captured_run_on_the_device(env, golden_dir):
    api, lib, ob = env
    mont = ob.fr_from_int

    class DevShift:
        def __init__(self, w, ro, rp, sg):
            self.shadow = ob.Stage3ShiftProver(w, ro, rp, sg)  # table construction only (init, transition)
            self.g = np.stack([mont(x) for x in sg])
            self.dev = api.ShiftSumcheckRounds(self.shadow.phase1_tables())

        def computeRoundEvals(self, claim):
            return self.dev.computeRoundEvals(claim)

        def bind(self, r):
            was2 = self.shadow.phase2
            self.dev.bind(mont(r))
            self.shadow.bind(r)
            if self.shadow.phase2 and not was2:  # the device's folded P / Q are the shadow's, then the second phase opens
                for got, want in zip(self.dev.tables(), [self.shadow.P[0], self.shadow.Q[0], self.shadow.P[1], self.shadow.Q[1], self.shadow.P[2],
                                                          self.shadow.Q[2], self.shadow.P[3], self.shadow.Q[3]]):
                    assert ob.fr_to_int(got[0]) == want[0]
                self.dev.deinit()
                self.dev = api.ShiftSumcheckRounds(self.shadow.phase2_tables(), phase2=True, gamma_powers=self.g)

        def finalClaims(self):
            t = self.dev.tables()
            self.dev.deinit()
            return dict(zip(("unexpanded_pc", "pc", "is_virtual", "is_first_in_sequence", "is_noop"), (x[0] for x in t[2:])))

    class DevInstr:
        def __init__(self, w, wm, ro, rp, ig):
            cols = [np.ascontiguousarray(wm[:, ob.R1CS_INPUT_NAMES.index(c)]) for c in ob.Stage3InstructionInputProver.COLS]
            eqs = [lib.fr_eq_table(np.stack([mont(x) for x in r])) for r in (ro, rp)]  # device-built eq tables
            self.dev = api.InstructionInputProver(cols + eqs, mont(ig))

        def computeRoundEvals(self, claim):
            return self.dev.computeRoundEvals(claim)

        def bind(self, r):
            self.dev.bind(mont(r))

    class DevRegisters:
        def __init__(self, w, wm, ro, rg):
            self.shadow = ob.Stage3RegistersProver(w, ro, rg)
            self.g = mont(rg)
            self.dev = api.RegistersClaimReductionRounds([ob._s3_tab(self.shadow.P), ob._s3_tab(self.shadow.Q)], self.g)
            # the three witness tables are folded alongside during phase 1 (stage3_prover.zig:2404-2416): a session that is only bound
            self.wit = lib.ProductSumcheckSession.open([np.ascontiguousarray(wm[:, ob.R1CS_INPUT_NAMES.index(c)]) for c in ("RdWriteValue", "Rs1Value", "Rs2Value")])
            self.phase2 = False

        def computeRoundEvals(self, claim):
            return self.dev.computeRoundEvals(claim)

        def bind(self, r):
            self.dev.bind(mont(r))
            self.shadow.bind(r)
            if self.phase2:
                return
            self.wit.bind(mont(r))
            if self.shadow.phase2:
                self.phase2 = True
                wit = [self.wit.read(j) for j in range(3)]  # folded on the device
                self.wit.close()
                self.dev.deinit()
                self.dev = api.RegistersClaimReductionRounds([ob._s3_tab(self.shadow.eq2)] + wit, self.g, phase2=True)

        def finalClaims(self):
            f = self.dev.finalClaims()
            self.dev.deinit()
            return f

    made = {}

    def make(w, wm, ro, rp, sg, ig, rg):
        made["i"] = (DevShift(w, ro, rp, sg), DevInstr(w, wm, ro, rp, ig), DevRegisters(w, wm, ro, rg))
        return made["i"]

    H.check_stage3_of_the_captured_run(make, golden_dir)
    made["i"][1].dev.deinit()


def test_stage3_provers_as_a_whole_on_the_device(env, golden_dir):
    """api.Stage3Prover builds the three instances itself — v and gamma^4 (1 - noop) as affine maps of the witness rows, the Q tables as
    weighted column sums (zg_fr_weighted_colsum), the eq / eq+1 prefix and suffix tables, the transitions — and runs the captured
    Stage 3 from the ELF to the printed bytes: input claims, eight compressed round polynomials, claims, final openings."""
    api, lib, ob = env
    P = ob._R_P
    le = lambda h: int.from_bytes(bytes.fromhex(h), "little")
    mont = ob.fr_from_int
    s3, w, wm, r_outer, r_product = H.stage3_inputs_of_the_captured_run(golden_dir)
    g = int(s3["shift_gamma_be"], 16)
    shift_g = np.stack([mont(pow(g, i, P)) for i in range(5)])
    instr_g, reg_g = mont(int(s3["instr_gamma_be"], 16)), mont(int(s3["reg_gamma_be"], 16))
    ro, rp = np.stack([mont(x) for x in r_outer]), np.stack([mont(x) for x in r_product])
    names = ("NextUnexpandedPC", "NextPC", "NextIsVirtual", "NextIsFirstInSequence", "LeftInstructionInput", "RightInstructionInput", "RdWriteValue",
             "Rs1Value", "Rs2Value")
    at_o = lib.fr_rows_mle(wm, ro)  # the device's own evaluator of the witness columns at a point
    at_p = lib.fr_rows_mle(wm, rp)
    idx = ob.R1CS_INPUT_NAMES.index
    noop = wm[:, idx("FlagIsNoop")]
    next_noop = np.concatenate([noop[1:], mont(1).reshape(1, 4)])
    outer = {n: at_o[idx(n)] for n in names}
    product = {"NextIsNoop": lib.fr_rows_mle(np.ascontiguousarray(next_noop).reshape(-1, 1, 4), rp)[0],
               "LeftInstructionInput": at_p[idx("LeftInstructionInput")], "RightInstructionInput": at_p[idx("RightInstructionInput")]}
    claims = api.Stage3Prover.inputClaims(outer, product, shift_g, instr_g, reg_g)
    assert [ob.fr_to_int(c) for c in claims] == [le(h) for h in s3["input_claims"]]
    p = api.Stage3Prover(wm, ro, rp, shift_g, instr_g, reg_g, claims, np.stack([mont(int(h, 16)) for h in s3["batching_coeffs_be"]]))
    try:
        for k, r in enumerate(s3["rounds"]):
            comp = p.computeRoundPolynomial()
            assert p.round_evals[0][:2] == [le(r["shift_p0"]), le(r["shift_p1"])], k
            assert [ob.fr_to_int(c) for c in comp] == [le(r["c0"]), le(r["c2"]), le(r["c3"])], k
            p.bindChallenge(mont(le(r["challenge"])))
            assert p.combined_claim == le(r["next_claim"]), k
        f = s3["final"]
        assert p.claims[1] == le(f["current_instr_claim"]) and p.claims[2] == le(f["current_reg_claim"])
        sh, rg = p.shift.finalClaims(), p.reg.finalClaims()
        assert [ob.fr_to_int(sh[k]) for k in ("unexpanded_pc", "pc", "is_noop")] == [le(f["shift_unexpanded_pc"]), le(f["shift_pc"]), le(f["shift_is_noop"])]
        assert [ob.fr_to_int(rg[k]) for k in ("rd_write_value", "rs1_value", "rs2_value")] == [le(f["reg_rd_write_value"]), le(f["reg_rs1_value"]), le(f["reg_rs2_value"])]
    finally:
        p.deinit()


@pytest.mark.parametrize("n,T", [(2, 4), (4, 16), (7, 128), (12, 4096), (13, 8192)])
def test_stage3_provers_against_the_restatement(env, n, T):
    """random witnesses (every column random: the provers are linear in them), T = 2^n padded cycles as the reference passes them — odd n
    (prefix one variable longer than the suffix), the transition after prefix_vars rounds: every round's evaluations of the device provers
    equal the restatement's, and so do the final claims"""
    api, lib, ob = env
    from tests import util as U
    rnd = lambda seed, k: ob.f_to_mont(ob.FR, U.random_raw256(seed, k))
    wm = rnd(9000 + n, T * 43).reshape(T, 43, 4)
    w = [[ob.fr_to_int(x) for x in row] for row in wm]
    ro, rp, ch = rnd(9100 + n, n), rnd(9200 + n, n), rnd(9300 + n, n)
    g = rnd(9400 + n, 3)
    gi = [ob.fr_to_int(x) for x in g]
    shift_g = [pow(gi[0], i, ob._R_P) for i in range(5)]
    roi, rpi = [ob.fr_to_int(x) for x in ro], [ob.fr_to_int(x) for x in rp]
    want = (ob.Stage3ShiftProver(w, roi, rpi, shift_g), ob.Stage3RegistersProver(w, roi, gi[2]))
    got = (api.ShiftPrefixSuffixProver(wm, ro, rp, np.stack([ob.fr_from_int(x) for x in shift_g])), api.RegistersPrefixSuffixProver(wm, ro, g[2]))
    claim = rnd(9500 + n, 1)[0]
    try:
        for k in range(n):
            for a, b in zip(got, want):
                assert np.array_equal(a.computeRoundEvals(claim), b.computeRoundEvals(claim)), (k, type(a).__name__)
                a.bind(ch[k])
                b.bind(ob.fr_to_int(ch[k]))
        for a, b in zip(got, want):
            fa, fb = a.finalClaims(), b.finalClaims()
            assert {k: ob.fr_to_int(v) for k, v in fa.items()} == {k: v for k, v in fb.items() if k in fa}
    finally:
        for a in got:
            a.deinit()


@pytest.mark.parametrize("rows,cols,m", [(1, 1, 1), (3, 5, 2), (16, 16, 4), (1024, 64, 3), (64, 1024, 4), (2048, 2048, 2)])
def test_weighted_colsum(env, rows, cols, m):
    """zg_fr_weighted_colsum against the plain sum, host and device entry points (one slab and many)"""
    api, lib, ob = env
    from tests import util as U
    tab = ob.f_to_mont(ob.FR, U.random_raw256(9600 + rows, rows * cols))
    wts = ob.f_to_mont(ob.FR, U.random_raw256(9700 + cols, m * rows)).reshape(m, rows, 4)
    got = lib.fr_weighted_colsum(tab, rows, cols, wts)
    t3 = tab.reshape(rows, cols, 4)
    for k in range(m):
        for c in sorted({0, cols // 2, cols - 1}):
            want = ob._fsum(ob._fmul(np.ascontiguousarray(t3[:, c]), np.ascontiguousarray(wts[k])))
            assert np.array_equal(got[k, c], want), (k, c)
    if rows * cols <= 4096:  # every entry at the small sizes
        for k in range(m):
            for c in range(cols):
                assert np.array_equal(got[k, c], ob._fsum(ob._fmul(np.ascontiguousarray(t3[:, c]), np.ascontiguousarray(wts[k]))))


def test_stage3_provers_at_full_size(env):
    """2^20 padded cycles (1.4 GB of witnesses): the restatement's Python loops do not reach this size, so the check is the protocol's own
    identities. Phase 1 of ShiftSumcheck computes p(0) AND p(1) from the tables: p(0) + p(1) must be the running claim in every one of its
    ten rounds, starting from the total sum. In the second phase p(1) is derived, so the proof of the pudding is the end: the last claim
    must equal  eq+1(r_outer, r) * v(r) + gamma^4 * eq+1(r_product, r) * (1 - noop(r))  resp.  eq(r_spartan, r) * (rd + g rs1 + g^2 rs2)(r)
    with the openings v(r), noop(r), rd(r) ... taken from an INDEPENDENT evaluator of the witness columns (zg_fr_rows_mle) at the
    challenge point — which also pins the provers' final claims."""
    api, lib, ob = env
    from tests import util as U
    n, P = 20, ob._R_P
    T = 1 << n
    rng = np.random.default_rng(20)
    wm = np.zeros((T, 43, 4), dtype=np.uint64)
    wm[:, :, 0] = rng.integers(0, 1 << 62, size=(T, 43), dtype=np.uint64)  # small canonical values, written as Montgomery limbs below
    cols = [api._I[c] for c in ("UnexpandedPC", "PC", "FlagVirtualInstruction", "FlagIsFirstInSequence", "FlagIsNoop", "RdWriteValue", "Rs1Value", "Rs2Value")]
    for c in cols:  # only the columns Stage 3 reads need to be field elements in Montgomery form
        wm[:, c] = lib.field_op(lib.FR, lib.OP_TO_MONT, np.ascontiguousarray(wm[:, c]))
    rnd = lambda seed, k: ob.f_to_mont(ob.FR, U.random_raw256(seed, k))
    ro, rp, ch, g = rnd(1, n), rnd(2, n), rnd(3, n), rnd(4, 2)
    gi = ob.fr_to_int(g[0])
    shift_g = np.stack([ob.fr_from_int(pow(gi, i, P)) for i in range(5)])
    sh = api.ShiftPrefixSuffixProver(wm, ro, rp, shift_g)
    rg = api.RegistersPrefixSuffixProver(wm, ro, g[1])
    to_int = ob.fr_to_int
    at = lambda cs, r: sum(c * pow(r, i, P) for i, c in enumerate(cs)) % P
    try:
        ev = [to_int(x) for x in sh.computeRoundEvals(ob.fr_from_int(0))]
        claim_s = (ev[0] + ev[1]) % P  # the total sum
        # registers: the total is the inner product of the two prefix tables
        claim_r = None
        for k in range(n):
            ev = [to_int(x) for x in sh.computeRoundEvals(ob.fr_from_int(claim_s))]
            assert (ev[0] + ev[1]) % P == claim_s, k
            er = [to_int(x) for x in rg.computeRoundEvals(ob.fr_from_int(claim_r if claim_r is not None else 0))]
            if claim_r is None:  # round 0: take the claim that makes p(1) = the true p(1): P . Q summed over the cube
                t = rg._rounds.tables()
                claim_r = to_int(ob._fsum(ob._fmul(t[0], t[1])))
                er = [to_int(x) for x in rg.computeRoundEvals(ob.fr_from_int(claim_r))]
            r = to_int(ch[k])
            claim_s = at(api.Stage3Prover.evalsToCoeffs(ev), r)
            claim_r = at(api.Stage3Prover.evalsToCoeffs(er), r)
            sh.bind(ch[k])
            rg.bind(ch[k])
        r_be = ch[::-1]  # the challenges as an MLE point (the first one bound the lowest index bit)
        opened = lib.fr_rows_mle(wm, r_be)
        fs, fr_ = sh.finalClaims(), rg.finalClaims()
        names = dict(unexpanded_pc="UnexpandedPC", pc="PC", is_virtual="FlagVirtualInstruction", is_first_in_sequence="FlagIsFirstInSequence",
                     is_noop="FlagIsNoop", rd_write_value="RdWriteValue", rs1_value="Rs1Value", rs2_value="Rs2Value")
        for k, v in list(fs.items()) + list(fr_.items()):
            assert np.array_equal(v, opened[api._I[names[k]]]), k
        o = {k: to_int(opened[api._I[v]]) for k, v in names.items()}
        gp = [pow(gi, i, P) for i in range(5)]
        v_r = (o["unexpanded_pc"] + gp[1] * o["pc"] + gp[2] * o["is_virtual"] + gp[3] * o["is_first_in_sequence"]) % P
        e1o, e1p = to_int(api.EqPlusOnePolynomial.mle(ro, r_be)), to_int(api.EqPlusOnePolynomial.mle(rp, r_be))
        assert claim_s == (e1o * v_r + gp[4] * e1p % P * (1 - o["is_noop"])) % P
        eq = 1
        for a, b in zip((to_int(x) for x in ro), (to_int(x) for x in r_be)):
            eq = eq * ((a * b + (1 - a) * (1 - b)) % P) % P
        g2 = to_int(g[1])
        assert claim_r == eq * (o["rd_write_value"] + g2 * o["rs1_value"] + g2 * g2 * o["rs2_value"]) % P
    finally:
        sh.deinit()
        rg.deinit()
