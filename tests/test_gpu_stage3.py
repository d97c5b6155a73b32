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
