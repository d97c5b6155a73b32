"""provers.py — the prover fold sites of stages 1-6 driven by the host transcript.

Part of the zolt_amd.api package (the host mirror of the reference's module API over libzolt_gpu.so); import zolt_amd.api,
which re-exports every name of every part."""
import numpy as np

from .. import lib
from ._base import *  # noqa: F401,F403
from .msm import *  # noqa: F401,F403
from .commitment import *  # noqa: F401,F403
from .wire import *  # noqa: F401,F403
from .poly import *  # noqa: F401,F403
from .sumcheck import *  # noqa: F401,F403
from .transcript import *  # noqa: F401,F403
from .witness import *  # noqa: F401,F403

def proveStage1(combined_poly, num_rounds, transcript):
    """The Stage-1 (outer Spartan) round loop of MultiStageProver.proveStage1 (src/zkvm/prover.zig:397-432) over
    JoltSpartanInterface.computeRoundPolynomial / bindChallenge (src/zkvm/r1cs/jolt_r1cs.zig:413-486): the combined polynomial
    lives in a LOW_PAIR device session; per round [p0, p1, 2 p1 - p0] comes back (64 bytes), is absorbed as round_poly_0/1/2,
    the challenge is challengeScalar("spartan_round") and goes back in for the fold. -> dict(round_polys, challenges, final_eval)."""
    poly = np.ascontiguousarray(combined_poly, dtype=np.uint64).reshape(-1, 4)
    n = poly.shape[0]
    sess = lib.SumcheckSession.open(poly, lib.SC_LOW_PAIR) if n >= 1 else None
    zero = np.zeros(4, dtype=np.uint64)
    round_polys, challenges = [], []
    try:
        for _ in range(num_rounds):
            cur = len(sess) if sess is not None else 0
            if cur <= 1:  # :421-430: single element -> [poly[0] or 0, 0, 0], bindChallenge does not fold (:462-465)
                p0 = sess.final() if cur == 1 else zero
                p1 = p2 = zero
            else:
                p0, p1 = sess.round_sums()  # even / odd sums (:436-444)
                p2 = _fr_sub(_fr_add(p1, p1), p0)  # :449
            round_polys.append(np.stack([p0, p1, p2]))
            transcript.appendScalar(b"round_poly_0", p0)
            transcript.appendScalar(b"round_poly_1", p1)
            transcript.appendScalar(b"round_poly_2", p2)
            ch = transcript.challengeScalar(b"spartan_round")
            challenges.append(ch)
            if cur > 1:
                sess.bind(ch)
        if sess is None:
            final = zero
        else:
            final = sess.final() if len(sess) == 1 else sess.read()[0]  # getFinalEval = combined_poly[0] (:492-497)
        return {"round_polys": np.array(round_polys).reshape(-1, 3, 4), "challenges": np.array(challenges).reshape(-1, 4), "final_eval": final}
    finally:
        if sess is not None:
            sess.close()


def computeRegEq(r, reg):
    """computeRegEq (src/zkvm/prover.zig:961-972): prod_i (bit_i(reg) ? r[i] : 1 - r[i])"""
    acc = 1
    for i, ri in enumerate(np.ascontiguousarray(r, dtype=np.uint64).reshape(-1, 4)):
        v = fr_to_int(ri)
        acc = acc * (v if (reg >> i) & 1 else 1 - v) % R_MOD
    return fr_from_int(acc)


def _highHalfRounds(evals, num_rounds, transcript, label):
    """the round loop of stages 5 and 6 (src/zkvm/prover.zig:902-944, 1055-1097) over a HIGH_HALF device session: p(0), p(1) = the sums
    of the two halves (64 bytes back per round), the proof keeps [p(0), 2 p(1) - p(0)], the challenge folds f[j] = (1 - r) f[j] + r f[j + half]
    on the device, claim = (1 - r) p(0) + r p(1) on the host"""
    if isinstance(evals, lib.SumcheckSession):  # a session built on the device (open_column)
        sess = evals
    else:
        ev = np.ascontiguousarray(evals, dtype=np.uint64).reshape(-1, 4)
        sess = lib.SumcheckSession.open(ev, lib.SC_HIGH_HALF)
    polys, chals, claims, initial = [], [], [], None
    try:
        for _ in range(num_rounds):
            p0, p1 = sess.round_sums()
            if initial is None:
                initial = _fr_add(p0, p1)  # the sum of the whole table: the stage's initial claim
            polys.append(np.stack([p0, _fr_sub(_fr_add(p1, p1), p0)]))
            ch = transcript.challengeScalar(label)
            chals.append(ch)
            sess.bind(ch)
            a, b, c = fr_to_int(p0), fr_to_int(p1), fr_to_int(ch)
            claims.append(fr_from_int(((1 - c) * a + c * b) % R_MOD))
        final = sess.final() if len(sess) == 1 else sess.read()[0]
    finally:
        sess.close()
    z = np.zeros((0, 4), dtype=np.uint64)
    return (np.stack(polys) if polys else np.zeros((0, 2, 4), dtype=np.uint64), np.stack(chals) if chals else z,
            np.stack(claims) if claims else z, final, final.copy() if initial is None else initial)


def valEvaluationTables(accesses, initial_ram, trace_len, k, r_address, r_cycle, start_address):
    """What ValEvaluationProver.init tabulates (src/zkvm/ram/val_evaluation.zig:423-470) -> (inc, wa, lt), (n, 4) each with
    n = ceilPow2(max(trace_len, 1)): inc from the writes of the trace (IncPolynomial.fromTrace, :92-165: host integers, one field element per
    write); wa[j] = eq(r_address, address written in cycle j) (WaPolynomial, :208-262) = a gather from the device's eq table of the
    reversed point; lt = LtPolynomial over the cube (:289-330, zg_fr_lt_table). accesses: [(timestamp, address, is_write, value)]."""
    r_address = np.ascontiguousarray(r_address, dtype=np.uint64).reshape(-1, 4)
    r_cycle = np.ascontiguousarray(r_cycle, dtype=np.uint64).reshape(-1, 4)
    n = 1
    while n < max(trace_len, 1):
        n <<= 1
    inc = np.zeros((n, 4), dtype=np.uint64)
    wa = np.zeros((n, 4), dtype=np.uint64)
    last = {}
    for addr, val in (initial_ram or {}).items():
        if addr >= start_address and (addr - start_address) // 8 < k:
            last[addr] = val
    eq = lib.fr_eq_table(np.ascontiguousarray(r_address[::-1])) if r_address.shape[0] else fr_from_int(1).reshape(1, 4)  # index bit i <-> r_address[i]
    for ts, addr, is_write, value in accesses:
        if not is_write or addr < start_address or (addr - start_address) // 8 >= k or ts >= trace_len:
            continue
        inc[ts] = fr_from_int((value - last.get(addr, 0)) % R_MOD)
        last[addr] = value
        wa[ts] = eq[((addr - start_address) // 8) % eq.shape[0]]
    full = lib.fr_lt_table(r_cycle)
    lt = full[:n] if full.shape[0] >= n else np.tile(full, (n // full.shape[0], 1))
    return inc, wa, np.ascontiguousarray(lt)


def proveStage4(accesses, initial_ram, trace_len, log_k, log_t, start_address, transcript):
    """MultiStageProver.proveStage4 (src/zkvm/prover.zig:713-828), Val evaluation: log_k "r_address" and log_t "r_cycle_val" challenges, the
    ValEvaluationProver over the memory trace (init_eval = 0), its initial claim, log2_ceil(trace_len) cubic rounds under "val_eval_round"
    on one device session, the final claim."""
    r_address = [transcript.challengeScalar(b"r_address") for _ in range(log_k)]
    r_cycle = [transcript.challengeScalar(b"r_cycle_val") for _ in range(log_t)]
    out = {"r_address": r_address, "r_cycle": r_cycle}
    if trace_len == 0:  # :764-768
        return out
    inc, wa, lt = valEvaluationTables(accesses, initial_ram, trace_len, 1 << log_k, np.array(r_address).reshape(-1, 4), np.array(r_cycle).reshape(-1, 4), start_address)
    pr = ValEvaluationProver(inc, wa, lt, np.zeros(4, dtype=np.uint64))
    first = pr.computeRoundPolynomial()  # the initial claim = p(0) + p(1) of the first round (a single entry: the product); kept for round 0
    pr.current_claim = _fr_add(first[0], first[1]) if inc.shape[0] >= 2 else first[0].copy()
    out["initial_claim"] = pr.computeInitialClaim()
    num_rounds = 0 if trace_len <= 1 else (trace_len - 1).bit_length()
    polys, chals = [], []
    for rd in range(num_rounds):
        rp = first if rd == 0 else pr.computeRoundPolynomial()
        polys.append(rp)
        ch = transcript.challengeScalar(b"val_eval_round")
        chals.append(ch)
        pr.bindChallengeWithPoly(ch, rp)
    f = pr.getFinalClaims()
    out.update(round_polys=polys, challenges=chals, final_openings=f,
               final_claim=fr_from_int(fr_to_int(f[0]) * fr_to_int(f[1]) % R_MOD * fr_to_int(f[2]) % R_MOD))
    pr.deinit()
    return out


def proveStage5(instructions, log_t, transcript):
    """MultiStageProver.proveStage5 (src/zkvm/prover.zig:829-958), register value evaluation: five r_register and log_t r_cycle_reg
    challenges, eq_evals[j] = eq(r_register, rd(j)) for the trace steps (a 32-entry table indexed by the rd field, zero past the trace),
    initial claim = the table's sum (the session's first pair of sums), then log2_ceil(trace_len) rounds on the device."""
    r_register = np.stack([transcript.challengeScalar(b"r_register") for _ in range(5)])
    r_cycle_reg = [transcript.challengeScalar(b"r_cycle_reg") for _ in range(log_t)]
    instr = np.asarray(instructions, dtype=np.uint32)
    if len(instr) == 0:  # :859-863
        return {"r_register": r_register, "r_cycle_reg": r_cycle_reg, "initial_claim": None}
    num_rounds = 0 if len(instr) <= 1 else (len(instr) - 1).bit_length()
    table = np.stack([computeRegEq(r_register, reg) for reg in range(32)])
    # eq_evals[j] = table[rd(j)], zero past the trace: one byte per cycle crosses, the lookup runs on the device (ZG_COL_LUT)
    rd = ((instr >> 7) & 31).astype(np.uint8)
    sess = lib.SumcheckSession.open_column((lib.COL_LUT, rd, 1, 32, table), len(rd), 1 << num_rounds, lib.SC_HIGH_HALF)
    polys, chals, claims, fin, initial = _highHalfRounds(sess, num_rounds, transcript, b"reg_eval_round")
    return {"r_register": r_register, "r_cycle_reg": r_cycle_reg, "initial_claim": initial, "round_polys": polys, "challenges": chals,
            "claims": claims, "final_claim": fin}


def proveStage6(trace_len, transcript):
    """MultiStageProver.proveStage6 (src/zkvm/prover.zig:990-1112), booleanity: the batching challenge, violation_evals = 0 for every step
    (the reference assumes a valid trace, :1024-1033) through the same round loop under "bool_round" """
    bool_challenge = transcript.challengeScalar(b"booleanity")
    if trace_len == 0:
        return {"bool_challenge": bool_challenge, "initial_claim": None}
    num_rounds = 0 if trace_len <= 1 else (trace_len - 1).bit_length()
    sess = lib.SumcheckSession.open_column((lib.COL_ZERO, None), 0, 1 << num_rounds, lib.SC_HIGH_HALF)  # cleared on the device, nothing uploaded
    polys, chals, claims, fin, initial = _highHalfRounds(sess, num_rounds, transcript, b"bool_round")
    return {"bool_challenge": bool_challenge, "initial_claim": initial, "round_polys": polys, "challenges": chals,
            "claims": claims, "final_claim": fin}


class RafEvaluationProver:
    """RafEvaluationProver (src/zkvm/ram/raf_checking.zig:262-470) with RaPolynomial's table in a LOW_PAIR device session: the
    cubic round polynomial's two sums are one kernel pass (zg_sumcheck_raf_round), the bind is the session's fold; the handful of
    scalar operations around them (base contribution, s(1), s(3), the Lagrange update of the claim) are host code as in the reference."""

    def __init__(self, ra_evals, start_address, log_k, initial_claim=None):
        """initial_claim None: the claim is computeInitialClaim() of the table (what prover.zig's Stage 2 passes in)"""
        ra = np.ascontiguousarray(ra_evals, dtype=np.uint64).reshape(-1, 4)
        assert ra.shape[0] == 1 << log_k
        self.sess = lib.SumcheckSession.open(ra, lib.SC_LOW_PAIR)
        self.start_address, self.log_k = int(start_address), log_k
        self.current_claim = self.computeInitialClaim() if initial_claim is None else np.array(initial_claim, dtype=np.uint64)
        self.bound_values = []
        self.round = 0
        self._final = ra[0].copy() if log_k == 0 else None

    def computeInitialClaim(self):
        """sum_k ra(k) * unmap(k) (:312-321), one pass over the resident table; only meaningful before the first bind"""
        return self.sess.raf_claim(self.start_address, 8)

    def computeRoundPolynomialCubic(self):
        """-> (4,4): s(0), s(1), s(2), s(3)   (:335-410)"""
        base = self.start_address % R_MOD
        power = 8
        for v in self.bound_values:
            base = (base + fr_to_int(v) * power) % R_MOD
            power *= 2
        s0, s2 = self.sess.raf_round(fr_from_int(base), power)
        a0, a2, claim = fr_to_int(s0), fr_to_int(s2), fr_to_int(self.current_claim)
        a1 = (claim - a0) % R_MOD
        a3 = (a0 - 3 * a1 + 3 * a2) % R_MOD
        return np.stack([s0, fr_from_int(a1), s2, fr_from_int(a3)])

    def updateClaim(self, evals, challenge):
        """Lagrange interpolation through evals at 0,1,2,3 evaluated at the challenge (:420-445)"""
        c = fr_to_int(challenge)
        e = [fr_to_int(x) for x in evals]
        inv = lambda x: pow(x % R_MOD, -1, R_MOD)
        L0 = (c - 1) * (c - 2) * (c - 3) * inv(-6)
        L1 = c * (c - 2) * (c - 3) * inv(2)
        L2 = c * (c - 1) * (c - 3) * inv(-2)
        L3 = c * (c - 1) * (c - 2) * inv(6)
        self.current_claim = fr_from_int((e[0] * L0 + e[1] * L1 + e[2] * L2 + e[3] * L3) % R_MOD)

    def bindChallenge(self, challenge):
        """RaPolynomial.bind (:162-174) + bookkeeping (:413-417)"""
        self.sess.bind(challenge)
        self.bound_values.append(np.array(challenge, dtype=np.uint64))
        self.round += 1

    def getFinalClaim(self):
        return self.sess.final() if len(self.sess) == 1 else self.sess.read()[0]

    def isComplete(self):
        return self.round >= self.log_k

    def deinit(self):
        self.sess.close()


def proveStage2(raf_prover, transcript):
    """The Stage-2 round loop (src/zkvm/prover.zig:523-548): cubic round polynomial, challengeScalar("raf_round"), claim
    update, bind. -> dict(round_polys [(s0, s2)], challenges, final_claim)."""
    polys, chals = [], []
    for _ in range(raf_prover.log_k):
        rp = raf_prover.computeRoundPolynomialCubic()
        polys.append(np.stack([rp[0], rp[2]]))  # the proof keeps s(0) and s(2) (:531-535)
        ch = transcript.challengeScalar(b"raf_round")
        chals.append(ch)
        raf_prover.updateClaim(rp, ch)
        raf_prover.bindChallenge(ch)
    return {"round_polys": np.array(polys).reshape(-1, 2, 4), "challenges": np.array(chals).reshape(-1, 4),
            "final_claim": raf_prover.getFinalClaim()}


class LassoAddressRounds:
    """The address phase of LassoProver.computeRoundPolynomial (src/zkvm/lasso/prover.zig:262-313): per round the eq values are
    summed by bit `round` of the u128 lookup index — eq_evals and indices stay resident in HBM for all LOG_K rounds."""

    def __init__(self, eq_evals, lookup_indices_u128):
        eq = np.ascontiguousarray(eq_evals, dtype=np.uint64).reshape(-1, 4)
        idx = np.ascontiguousarray(lookup_indices_u128, dtype=np.uint64).reshape(-1, 2)
        assert eq.shape[0] == idx.shape[0]
        self.n = eq.shape[0]
        self._eq = lib.DeviceBuffer.from_host(eq)
        self._idx = lib.DeviceBuffer.from_host(idx)

    def computeAddressRoundPoly(self, round_bit):
        """-> coeffs [sum_0, sum_1 - sum_0, 0] (:304-306)"""
        s0, s1 = lib.fr_bit_split_sums_dev(self._eq.ptr, self._idx.ptr, self.n, round_bit)
        return np.stack([s0, _fr_sub(s1, s0), np.zeros(4, dtype=np.uint64)])

    def deinit(self):
        self._eq.free()
        self._idx.free()


def cubicAtPoint(evals, challenge):
    """Lagrange interpolation through evals at 0,1,2,3, evaluated at the challenge — the claim update every cubic prover repeats
    (val_evaluation.zig:630-660, instruction_lookups.zig:250-270, product_remainder.zig:534-559)"""
    c = fr_to_int(challenge)
    e = [fr_to_int(x) for x in evals]
    inv = lambda x: pow(x % R_MOD, -1, R_MOD)
    L0 = (c - 1) * (c - 2) * (c - 3) * inv(-6)
    L1 = c * (c - 2) * (c - 3) * inv(2)
    L2 = c * (c - 1) * (c - 3) * inv(-2)
    L3 = c * (c - 1) * (c - 2) * inv(6)
    return fr_from_int((e[0] * L0 + e[1] * L1 + e[2] * L2 + e[3] * L3) % R_MOD)


def interpolateDegree3(evals):
    """UniPoly.interpolateDegree3 (src/poly/mod.zig:632-677): coefficients [c0, c1, c2, c3] from p(0), p(1), p(2), p(3)"""
    p0, p1, p2, p3 = (fr_to_int(x) for x in evals)
    inv6, inv2 = pow(6, -1, R_MOD), pow(2, -1, R_MOD)
    c1 = (-11 * p0 + 18 * p1 - 9 * p2 + 2 * p3) * inv6 % R_MOD
    c2 = (2 * p0 - 5 * p1 + 4 * p2 - p3) * inv2 % R_MOD
    c3 = (-p0 + 3 * p1 - 3 * p2 + p3) * inv6 % R_MOD
    return np.stack([fr_from_int(p0), fr_from_int(c1), fr_from_int(c2), fr_from_int(c3)])


def evalsToCompressed(evals):
    """UniPoly.evalsToCompressed (:682-685): [c0, c2, c3]"""
    c = interpolateDegree3(evals)
    return np.stack([c[0], c[2], c[3]])


class ValEvaluationProver:
    """ValEvaluationProver's sumcheck loop (src/zkvm/ram/val_evaluation.zig:545-700): inc * wa * lt over the cycles, LowToHigh.
    The three tables stay in one device session (zg_psc_*); the claim update is the reference's host algebra."""
    FACTORS = (0, 1, 2)

    def __init__(self, inc_evals, wa_evals, lt_evals, claim):
        tabs = [inc_evals, wa_evals] + ([] if lt_evals is None else [lt_evals])
        self._s = lib.ProductSumcheckSession.open(tabs)
        self.current_claim = np.ascontiguousarray(claim, dtype=np.uint64).copy()
        self.round = 0

    def computeInitialClaim(self):
        return self.current_claim.copy()

    def effectiveLen(self):
        return len(self._s)

    def computeRoundPolynomial(self):
        """[p(0), p(1), p(2), p(3)] (:554-603)"""
        if len(self._s) < 2:  # :559-565: a single entry: p(0) = the product, the rest zero
            out = np.zeros((4, 4), dtype=np.uint64)
            acc = 1
            for v in self._s.final():
                acc = acc * fr_to_int(v) % R_MOD
            out[0] = fr_from_int(acc)
            return out
        return self._s.round_evals(self.FACTORS)

    def bindChallengeWithPoly(self, r, round_poly):
        """:609-660: fold every table, claim = p(r) by cubic Lagrange"""
        if len(self._s) >= 2:
            self._s.bind(r)
            self.current_claim = cubicAtPoint(round_poly, r)
        self.round += 1

    def getFinalClaims(self):
        return list(self._s.final())

    def deinit(self):
        self._s.close()


class ValFinalProver(ValEvaluationProver):
    """ValFinalProver's loop (src/zkvm/ram/val_final.zig:144-230): inc * wa, the same cubic message format"""
    FACTORS = (0, 1)

    def __init__(self, inc_evals, wa_evals, claim):
        super().__init__(inc_evals, wa_evals, None, claim)


class OutputSumcheckProver:
    """OutputSumcheckProver's loop (src/zkvm/ram/output_check.zig:375-499): eq * io_mask * (val_final - val_io); val_init is folded
    alongside for the final claims. Tables 0..4 = eq_r_address, io_mask, val_final, val_io, val_init."""

    def __init__(self, eq_r_address, io_mask, val_final, val_io, val_init, claim):
        self._s = lib.ProductSumcheckSession.open([eq_r_address, io_mask, val_final, val_io, val_init])
        self.current_size = len(self._s)
        self.current_claim = np.ascontiguousarray(claim, dtype=np.uint64).copy()
        self._coeff = np.stack([fr_from_int(1), fr_from_int(R_MOD - 1)])  # vf - vio

    def roundEvals(self):
        """s(0), s(1), s(2), s(3) (:378-430)"""
        return self._s.round_evals((0, 1), (2, 3), self._coeff)

    def computeRoundPolynomial(self):
        """compressed coefficients [c0, c2, c3] (:445)"""
        return evalsToCompressed(self.roundEvals())

    def bindChallenge(self, r):
        self._s.bind(r)
        self.current_size //= 2

    def updateClaim(self, evals, challenge):
        """:482-499: c0 + c1 r + c2 r^2 + c3 r^3 with c2, c3 from lagrangeC2 / lagrangeC3"""
        c = interpolateDegree3(evals)
        r = fr_to_int(challenge)
        c0, c2, c3 = fr_to_int(c[0]), fr_to_int(c[2]), fr_to_int(c[3])
        c1 = (fr_to_int(evals[1]) - c0 - c2 - c3) % R_MOD
        self.current_claim = fr_from_int((c0 + c1 * r + c2 * r * r + c3 * r * r * r) % R_MOD)

    def getFinalClaims(self):
        f = self._s.final()
        return {"val_final": f[2], "val_init": f[4], "val_io": f[3], "eq_r_address": f[0], "io_mask": f[1]}

    def deinit(self):
        self._s.close()


class InstructionLookupsClaimReductionProver:
    """InstructionLookupsClaimReductionProver's loop (src/zkvm/claim_reductions/instruction_lookups.zig:146-284):
    eq * (lookup_output + gamma left + gamma^2 right). Tables 0..3 = eq_evals, lookup_outputs, left_operands, right_operands."""

    def __init__(self, eq_evals, lookup_outputs, left_operands, right_operands, gamma, claim):
        self._s = lib.ProductSumcheckSession.open([eq_evals, lookup_outputs, left_operands, right_operands])
        g = fr_to_int(gamma)
        self._coeff = np.stack([fr_from_int(1), fr_from_int(g), fr_from_int(g * g % R_MOD)])
        self._s.set_points(0b0101)  # only s(0) and s(2) are read from the tables
        self.current_claim = np.ascontiguousarray(claim, dtype=np.uint64).copy()
        self.round = 0

    def computeRoundPolynomialCubic(self):
        """[s0, s1, s2, s3]: s0, s2 from the tables, s1 = claim - s0, s3 = s0 - 3 s1 + 3 s2 (:146-200)"""
        ev = self._s.round_evals((0,), (1, 2, 3), self._coeff)
        s0, s2 = fr_to_int(ev[0]), fr_to_int(ev[2])
        s1 = (fr_to_int(self.current_claim) - s0) % R_MOD
        s3 = (s0 - 3 * s1 + 3 * s2) % R_MOD
        return np.stack([ev[0], fr_from_int(s1), ev[2], fr_from_int(s3)])

    def bindChallenge(self, challenge):
        self._s.bind(challenge)
        self.round += 1

    def updateClaim(self, evals, challenge):
        self.current_claim = cubicAtPoint(evals, challenge)

    def getOpeningClaims(self):
        f = self._s.final()
        return {"lookup_output": f[1], "left_operand": f[2], "right_operand": f[3]}

    def deinit(self):
        self._s.close()


class ProductVirtualRemainderProver:
    """ProductVirtualRemainderProver's loop (src/zkvm/spartan/product_remainder.zig:269-394) over the fused left / right tables:
    Gruen's (t0, t_inf) on the device under the split-eq weights (prefix tables resident in HBM), the cubic from
    GruenSplitEqPolynomial.computeCubicRoundPoly on the host."""

    def __init__(self, left_evals, right_evals, tau_low, lagrange_kernel, uni_skip_claim):
        self._s = lib.ProductSumcheckSession.open([left_evals, right_evals])
        self.split_eq = GruenSplitEqPolynomial(tau_low, lagrange_kernel)
        self.current_claim = np.ascontiguousarray(uni_skip_claim, dtype=np.uint64).copy()
        self.current_round = 0

    def roundEvals(self):
        if len(self._s) < 2:
            return None
        d_out, n_out, d_in, n_in = self.split_eq.getWindowEqTablesDev(1)
        t0, t_inf = self._s.round_gruen((0, 1), d_out, n_out, d_in, n_in)
        return self.split_eq.computeCubicRoundPoly(t0, t_inf, self.current_claim)

    def computeRoundPolynomial(self):
        """compressed [c0, c2, c3]; [claim, 0, 0] without groups (:274-276)"""
        ev = self.roundEvals()
        if ev is None:
            z = np.zeros(4, dtype=np.uint64)
            return np.stack([self.current_claim, z, z])
        return evalsToCompressed(ev)

    def bindChallenge(self, challenge):
        self._s.bind(challenge)
        self.split_eq.bind(challenge)
        self.current_round += 1

    def updateClaim(self, round_evals, challenge):
        self.current_claim = cubicAtPoint(round_evals, challenge)

    def getFinalClaim(self):
        f = self._s.final()
        return fr_from_int(fr_to_int(f[0]) * fr_to_int(f[1]) % R_MOD)

    def deinit(self):
        self._s.close()
        self.split_eq.deinit()


PRODUCT_VIRTUAL_TARGETS = None  # filled below (uniskipTargets is defined later in this module)
PRODUCT_VIRTUAL_COEFFS_PER_J = None


def _product_fused_rows(weights):
    """fusedLeft / fusedRight (src/zkvm/spartan/product_remainder.zig:113-134, extractProductInputs :436-476) as two affine maps of a WINDOW of
    two consecutive cycles' inputs (86 columns + the constant): left = w0 LeftInput + (w1 + w2) IsRdNotZero + w3 LookupOutput + w4 Jump,
    right = w0 RightInput + w1 WLFlag + w2 Jump + w3 Branch + w4 (1 - IsNoop of the NEXT cycle). weights: five integers mod r."""
    W = 2 * NUM_R1CS_INPUTS + 1
    left, right = [0] * W, [0] * W
    w0, w1, w2, w3, w4 = [int(x) % R_MOD for x in weights]
    left[_I["LeftInstructionInput"]] = w0
    left[_I["FlagIsRdNotZero"]] = (w1 + w2) % R_MOD
    left[_I["LookupOutput"]] = w3
    left[_I["FlagJump"]] = w4
    right[_I["RightInstructionInput"]] = w0
    right[_I["FlagWriteLookupOutputToRD"]] = w1
    right[_I["FlagJump"]] = w2
    right[_I["FlagBranch"]] = w3
    right[NUM_R1CS_INPUTS + _I["FlagIsNoop"]] = (-w4) % R_MOD
    right[W - 1] = w4
    return left, right


def _witnesses_with_sentinel(cycle_witnesses):
    """the cycle-major matrix on the device with one more row after the last cycle whose IsNoop flag is set: the window of the last cycle
    reads NextIsNoop = 1 there (product_remainder.zig:468-474)"""
    w = np.ascontiguousarray(cycle_witnesses, dtype=np.uint64).reshape(-1, NUM_R1CS_INPUTS, 4)
    ext = np.zeros((w.shape[0] + 1, NUM_R1CS_INPUTS, 4), dtype=np.uint64)
    ext[:-1] = w
    ext[-1, _I["FlagIsNoop"]] = fr_from_int(1)
    return lib.DeviceBuffer.from_host(ext), w.shape[0]


def productVirtualExtendedEvals(cycle_witnesses, tau):
    """computeProductVirtualExtendedEvals (src/zkvm/r1cs/univariate_skip.zig:607-678): t1 at -3, 3, -4, 4 = sum_x eq(tau[0..log n), x) *
    fused_left(x) * fused_right(x) — one product-sum launch over the resident witnesses (two-cycle windows)"""
    d_rows, n = _witnesses_with_sentinel(cycle_witnesses)
    log_n = max(n - 1, 0).bit_length()
    tau = np.ascontiguousarray(tau, dtype=np.uint64).reshape(-1, 4)
    rows = []
    for coeffs in PRODUCT_VIRTUAL_COEFFS_PER_J:
        rows.extend(_product_fused_rows(coeffs))
    m = np.stack([np.stack([fr_from_int(v) for v in row]) for row in rows])
    d_w = lib.DeviceBuffer((1 << log_n) * 32)
    lib.fr_eq_table_dev(tau[:log_n], d_w.ptr)
    out = lib.fr_rows_affine_prodsum_dev(d_rows.ptr, n, 2 * NUM_R1CS_INPUTS, m, 4, d_w.ptr, 1, stride=NUM_R1CS_INPUTS)
    d_w.free()
    d_rows.free()
    return out


def buildUniskipFirstRoundPoly(domain_size, degree, base_evals, extended_evals, tau_high):
    """buildUniskipFirstRoundPoly (univariate_skip.zig:486-546) -> 3 * degree + 1 coefficients of s1 = L(tau_high, .) * t1 (host integers)"""
    targets = uniskipTargets(domain_size, degree)
    t1 = [0] * (2 * degree + 1)
    base_left = -((domain_size - 1) // 2)
    if base_evals is not None:
        for i, v in enumerate(base_evals):
            t1[base_left + i + degree] = fr_to_int(v)
    for z, v in zip(targets, extended_evals):
        t1[z + degree] = fr_to_int(v)
    t1c = interpolateIntDomain(t1, -degree)
    lagc = interpolateIntDomain([fr_to_int(x) for x in lagrangeEvals(tau_high, domain_size)], base_left)
    s1 = [0] * (3 * degree + 1)
    for i, a in enumerate(lagc):
        for j, b in enumerate(t1c):
            if i + j < len(s1):
                s1[i + j] = (s1[i + j] + a * b) % R_MOD
    return np.stack([fr_from_int(v) for v in s1])


def productVirtualRemainderProverFromWitnesses(cycle_witnesses, r0, tau, uni_skip_claim):
    """ProductVirtualRemainderProver.init (product_remainder.zig:166-243): the fused left / right tables built ON THE DEVICE from the resident
    witnesses (two affine maps of a two-cycle window, zg_fr_rows_affine_dev) straight into the prover's product session"""
    d_rows, n = _witnesses_with_sentinel(cycle_witnesses)
    padded = 1
    while padded < n:
        padded *= 2
    tau = np.ascontiguousarray(tau, dtype=np.uint64).reshape(-1, 4)
    left, right = _product_fused_rows([fr_to_int(x) for x in lagrangeEvals(r0, 5)])
    m = np.stack([np.stack([fr_from_int(v) for v in row]) for row in (left, right)])
    d_l, d_r = lib.DeviceBuffer(padded * 32), lib.DeviceBuffer(padded * 32)
    lib.fr_rows_affine_dev(d_rows.ptr, n, 2 * NUM_R1CS_INPUTS, m, 2, 1, padded, [d_l.ptr, d_r.ptr], stride=NUM_R1CS_INPUTS)
    p = ProductVirtualRemainderProver.__new__(ProductVirtualRemainderProver)
    p._s = lib.ProductSumcheckSession.open_dev([d_l.ptr, d_r.ptr], padded)
    lib.sync()
    for b in (d_l, d_r, d_rows):
        b.free()
    p.split_eq = GruenSplitEqPolynomial(tau[:-1], lagrangeKernel(r0, tau[-1], 5))
    p.current_claim = np.ascontiguousarray(uni_skip_claim, dtype=np.uint64).copy()
    p.current_round = 0
    return p


class InstructionInputProver:
    """InstructionInputProver's loop (src/zkvm/spartan/stage3_prover.zig:2029-2150): ten cycle-length tables,
    f = (eq_outer + gamma^2 eq_product) * (right_is_rs2 * rs2 + right_is_imm * imm + gamma (left_is_rs1 * rs1 + left_is_pc * pc)) as four
    product terms of one multi-term round (zg_psc_round_expr). Table order: left_is_rs1, rs1_value, left_is_pc, unexpanded_pc,
    right_is_rs2, rs2_value, right_is_imm, imm, eq_outer, eq_product."""
    NAMES = ("left_is_rs1", "rs1_value", "left_is_pc", "unexpanded_pc", "right_is_rs2", "rs2_value", "right_is_imm", "imm", "eq_outer", "eq_product")

    def __init__(self, tables, gamma, d_tables=None, n=None):
        """tables: ten host arrays, or None with d_tables = ten device pointers of n entries each (copied into the session)"""
        if d_tables is None:
            assert len(tables) == 10
            self._s = lib.ProductSumcheckSession.open(tables)
        else:
            assert len(d_tables) == 10
            self._s = lib.ProductSumcheckSession.open_dev(d_tables, n)
        g = fr_to_int(gamma)
        w_right = np.stack([fr_from_int(1), fr_from_int(g * g % R_MOD)])  # eq_outer + gamma^2 eq_product
        w_left = np.stack([fr_from_int(g), fr_from_int(g * g * g % R_MOD)])  # gamma times the same weight
        # two pair-sum terms (ZG_PSC_PAIR_SUM): (is_rs2 rs2 + is_imm imm) under the right weight, (is_rs1 rs1 + is_pc pc) under gamma times it
        self._terms = [((4, 5, 6, 7), (8, 9), w_right, True), ((0, 1, 2, 3), (8, 9), w_left, True)]
        self._s.set_points(0b1101)  # p(1) comes from the claim
        self.current_size = len(self._s)

    def computeRoundEvals(self, previous_claim):
        """[p(0), previous_claim - p(0), p(2), p(3)] (:2029-2100)"""
        ev = self._s.round_expr(self._terms)
        return np.stack([ev[0], _fr_sub(previous_claim, ev[0]), ev[2], ev[3]])

    def bind(self, r_j):
        self._s.bind(r_j)
        self.current_size //= 2

    def finalClaims(self):
        return dict(zip(self.NAMES, self._s.final()))

    def deinit(self):
        self._s.close()


class ShiftSumcheckRounds:
    """The two round loops of ShiftSumcheckProver (src/zkvm/spartan/stage3_prover.zig:1343-1500,1782-1817) over given tables; the
    prefix-suffix construction of P / Q and the phase transition (:1504-1780) stay the caller's.
    phase 1: P_0_outer, Q_0_outer, P_1_outer, Q_1_outer, P_0_prod, Q_0_prod, P_1_prod, Q_1_prod (H = sum of the four P * Q);
    phase 2: eq+1_outer, eq+1_prod, unexpanded_pc, pc, is_virtual, is_first_in_sequence, is_noop and gamma_powers[0..4]."""

    def __init__(self, tables, phase2=False, gamma_powers=None):
        self.phase2 = phase2
        self._s = lib.ProductSumcheckSession.open(tables)
        if not phase2:
            assert len(tables) == 8
            self._terms = [((0, 1, 2, 3), (), None, True), ((4, 5, 6, 7), (), None, True)]  # two pair sums: P0 Q0 + P1 Q1 per opening
            self._s.set_points(0b0111)  # a quadratic: p(0), p(1), p(2)
        else:
            assert len(tables) == 7
            g = [fr_to_int(x) for x in np.ascontiguousarray(gamma_powers, dtype=np.uint64).reshape(-1, 4)]
            val = np.stack([fr_from_int(1), fr_from_int(g[1]), fr_from_int(g[2]), fr_from_int(g[3])])
            # gamma^4 (1 - noop) eq_prod = gamma^4 eq_prod - gamma^4 noop eq_prod: the same field value at every t
            self._terms = [((0,), (2, 3, 4, 5), val), ((), (1,), fr_from_int(g[4]).reshape(1, 4)), ((6,), (1,), fr_from_int((-g[4]) % R_MOD).reshape(1, 4))]
            self._s.set_points(0b0101)  # p(0) and p(2); p(1) comes from the claim

    def computeRoundEvals(self, previous_claim):
        """phase 1 (:1351-1392): [p(0), p(1), p(2)] all computed; phase 2 (:1399-1455): [p(0), previous_claim - p(0), p(2)]"""
        ev = self._s.round_expr(self._terms)
        if not self.phase2:
            return np.stack([ev[0], ev[1], ev[2]])
        return np.stack([ev[0], _fr_sub(previous_claim, ev[0]), ev[2]])

    def bind(self, r_j):
        """bindPhase1 (:1479-1503) / bindPhase2 (:1782-1817): every table by lo + r (hi - lo)"""
        self._s.bind(r_j)

    def tables(self):
        return [self._s.read(j) for j in range(self._s.tables())]

    def deinit(self):
        self._s.close()


class RegistersClaimReductionRounds:
    """The two round loops of RegistersClaimReductionProver (stage3_prover.zig:2326-2481) over given tables.
    phase 1: P, Q (prefix-sized; the witness tables rd_write_value / rs1_value / rs2_value the reference folds alongside, :2404-2416, have
    the full cycle length and live in a second session that is only bound); phase 2: eq, rd_write_value, rs1_value, rs2_value."""

    def __init__(self, tables, gamma, phase2=False):
        self.phase2 = phase2
        self._s = lib.ProductSumcheckSession.open(tables)
        g = fr_to_int(gamma)
        self._coeff = np.stack([fr_from_int(1), fr_from_int(g), fr_from_int(g * g % R_MOD)])
        self._s.set_points(0b0101)  # p(0) and p(2); p(1) comes from the claim

    def computeRoundEvals(self, previous_claim):
        """[p(0), previous_claim - p(0), p(2)] (:2334-2389)"""
        ev = self._s.round_evals((0,), (1, 2, 3), self._coeff) if self.phase2 else self._s.round_evals((0, 1))
        return np.stack([ev[0], _fr_sub(previous_claim, ev[0]), ev[2]])

    def bind(self, r_j):
        self._s.bind(r_j)

    def finalClaims(self):
        """:2483-2494 (phase 2)"""
        f = self._s.final()
        return {"rd_write_value": f[1], "rs1_value": f[2], "rs2_value": f[3]}

    def tables(self):
        return [self._s.read(j) for j in range(self._s.tables())]

    def deinit(self):
        self._s.close()


def _evaluate_mle_low(table, point):
    """evaluateMle of the Stage-3 provers (src/zkvm/spartan/stage3_prover.zig:1820-1838): the point's first entry binds the LOW index bit"""
    t = np.ascontiguousarray(table, dtype=np.uint64).reshape(-1, 4)
    point = np.ascontiguousarray(point, dtype=np.uint64).reshape(-1, 4)
    if t.shape[0] == (1 << point.shape[0]):  # the whole cube: one device call (zg_fr_dense_evaluate has the same bit order)
        return lib.fr_dense_evaluate(t, point)
    for r in point:
        if t.shape[0] == 1:
            break
        t = lib.fr_bind_low(t, r)
    return t[0]


def _witness_maps_dev(d_rows, n_rows, n_pad, maps):
    """cycle-length tables that are affine maps of the witness rows, built in HBM: maps = rows of NUM_R1CS_INPUTS + 1 canonical integers
    (the constant last; a plain column is a single 1) -> (DeviceBuffer holding len(maps) tables of n_pad entries, their device pointers).
    zg_fr_rows_affine_dev takes at most 16 maps per launch."""
    buf = lib.DeviceBuffer(len(maps) * n_pad * 32)
    ptrs = [buf.ptr + i * n_pad * 32 for i in range(len(maps))]
    for a in range(0, len(maps), 16):
        part = maps[a:a + 16]
        coeffs = np.stack([np.stack([fr_from_int(x) for x in row]) for row in part])
        lib.fr_rows_affine_dev(d_rows, n_rows, NUM_R1CS_INPUTS, coeffs, len(part), 1, n_pad, ptrs[a:a + 16])
    return buf, ptrs


def _column_map(name):
    m = [0] * (NUM_R1CS_INPUTS + 1)
    m[_I[name]] = 1
    return m


def _colsum_dev(d_table, rows, cols, weights):
    """zg_fr_weighted_colsum_dev with host weights in and the (m, cols, 4) sums out (sqrt(T)-sized both)"""
    weights = np.ascontiguousarray(weights, dtype=np.uint64)
    m = weights.size // (4 * rows)
    d_w, d_o = lib.DeviceBuffer.from_host(weights), lib.DeviceBuffer(m * cols * 32)
    lib.fr_weighted_colsum_dev(d_table, rows, cols, d_w.ptr, m, d_o.ptr)
    lib.sync()
    out = d_o.to_host().reshape(m, cols, 4)
    d_w.free()
    d_o.free()
    return out


class ShiftPrefixSuffixProver:
    """ShiftPrefixSuffixProver (src/zkvm/spartan/stage3_prover.zig:928-1919) as a whole, over the cycle witnesses.
    init (:979-1112): P_0 = eq+1(r_lo, .), P_1 = is_max(r_lo) at 0, suffixes eq(r_hi, .) / eq+1(r_hi, .) are sqrt(T)-sized device table builds
    (EqPlusOnePrefixSuffixPoly); the batched witness value v = upc + g pc + g^2 virt + g^3 first and g^4 (1 - noop) are AFFINE maps of a
    cycle's R1CS inputs (one zg_fr_rows_affine launch over the witness matrix), and the four Q tables are their column sums under the
    suffixes (zg_fr_weighted_colsum: the table read as 2^suffix_vars rows of 2^prefix_vars columns). Rounds: ShiftSumcheckRounds. The five
    witness columns are folded on the device from round 0 — low-to-high binds by the prefix challenges give exactly the
    sum_i eq(r_prefix, i) * witness[j * 2^prefix + i] the reference rebuilds at the transition (:1628-1664) — and the two eq+1 tables of the
    second phase are prefix evaluations times the suffix tables (:1560-1610), sqrt(T) entries."""
    COLS = ("UnexpandedPC", "PC", "FlagVirtualInstruction", "FlagIsFirstInSequence", "FlagIsNoop")

    def __init__(self, cycle_witnesses, r_outer, r_product, gamma_powers, d_rows=None):
        """cycle_witnesses: the padded trace's (2^n, 43, 4) R1CS inputs in host memory, or None with d_rows = the same matrix already in
        HBM (Stage 1's upload: StreamingOuterProver keeps it)"""
        r_outer = np.ascontiguousarray(r_outer, dtype=np.uint64).reshape(-1, 4)
        r_product = np.ascontiguousarray(r_product, dtype=np.uint64).reshape(-1, 4)
        n = r_outer.shape[0]
        assert n >= 2 and r_product.shape[0] == n
        own = None
        if d_rows is None:
            w = np.ascontiguousarray(cycle_witnesses, dtype=np.uint64).reshape(-1, NUM_R1CS_INPUTS, 4)
            assert w.shape[0] == (1 << n)  # the padded trace, as generateStage3Proof receives it
            own = lib.DeviceBuffer.from_host(w)
            d_rows = own.ptr
        self.gamma_powers = np.ascontiguousarray(gamma_powers, dtype=np.uint64).reshape(5, 4).copy()
        g = [fr_to_int(x) for x in self.gamma_powers]
        self.suffix_n_vars = n // 2
        self.prefix_n_vars = n - self.suffix_n_vars
        ps, ss = 1 << self.prefix_n_vars, 1 << self.suffix_n_vars
        self._eq = (EqPlusOnePrefixSuffixPoly(r_outer), EqPlusOnePrefixSuffixPoly(r_product))
        m = [[0] * (NUM_R1CS_INPUTS + 1) for _ in range(2)]
        for c, k in zip(self.COLS[:4], (1, g[1], g[2], g[3])):
            m[0][_I[c]] = k
        m[1][_I["FlagIsNoop"]], m[1][NUM_R1CS_INPUTS] = (-g[4]) % R_MOD, g[4]  # g^4 (1 - noop): the reference scales the sum, the same value
        N = 1 << n
        buf, ptrs = _witness_maps_dev(d_rows, N, N, m + [_column_map(c) for c in self.COLS])  # v, g^4 (1 - noop), the five columns
        q_o = _colsum_dev(ptrs[0], ss, ps, np.stack([self._eq[0].suffix_0, self._eq[0].suffix_1]))
        q_p = _colsum_dev(ptrs[1], ss, ps, np.stack([self._eq[1].suffix_0, self._eq[1].suffix_1]))
        self._rounds = ShiftSumcheckRounds([self._eq[0].prefix_0, q_o[0], self._eq[0].prefix_1, q_o[1], self._eq[1].prefix_0, q_p[0], self._eq[1].prefix_1, q_p[1]])
        self._wit = lib.ProductSumcheckSession.open_dev(ptrs[2:], N)
        lib.sync()
        buf.free()
        if own is not None:
            own.free()
        self.current_prefix_size, self.in_phase2, self.sumcheck_challenges = ps, False, []

    def computeRoundEvals(self, previous_claim):
        return self._rounds.computeRoundEvals(previous_claim)

    def bind(self, r_j):
        """:1458-1472"""
        r_j = np.ascontiguousarray(r_j, dtype=np.uint64).reshape(4)
        self._rounds.bind(r_j)
        if self.in_phase2:
            return
        transition = self.current_prefix_size == 2  # shouldTransitionToPhase2 (:1474-1477)
        self._wit.bind(r_j)
        self.sumcheck_challenges.append(r_j.copy())
        self.current_prefix_size //= 2
        if transition:
            self._transition()

    def _transition(self):
        """transitionToPhase2 (:1506-1700)"""
        tabs = []
        for e in self._eq:
            e0, e1 = _evaluate_mle_low(e.prefix_0, self.sumcheck_challenges), _evaluate_mle_low(e.prefix_1, self.sumcheck_challenges)
            ss = e.suffix_0.shape[0]
            tabs.append(lib.field_op(lib.FR, lib.OP_ADD, lib.field_op(lib.FR, lib.OP_MUL, np.tile(e0, (ss, 1)), e.suffix_0),
                                     lib.field_op(lib.FR, lib.OP_MUL, np.tile(e1, (ss, 1)), e.suffix_1)))
        wit = [self._wit.read(j) for j in range(5)]
        self._wit.close()
        self._wit = None
        self._rounds.deinit()
        self._rounds = ShiftSumcheckRounds(tabs + wit, phase2=True, gamma_powers=self.gamma_powers)
        self.in_phase2 = True

    def finalClaims(self):
        """:1860-1876"""
        t = self._rounds.tables()
        return dict(zip(("unexpanded_pc", "pc", "is_virtual", "is_first_in_sequence", "is_noop"), (x[0] for x in t[2:])))

    def deinit(self):
        if self._wit is not None:
            self._wit.close()
        self._rounds.deinit()


class RegistersPrefixSuffixProver:
    """RegistersPrefixSuffixProver (src/zkvm/spartan/stage3_prover.zig:2156-2495) as a whole: P = eq(r_lo, .), Q = column sums of
    rd + g rs1 + g^2 rs2 (an affine map of the witness rows) under eq(r_hi, .); the three witness tables are folded on the device from
    round 0, as the reference folds them (:2404-2416); the second phase's eq table is eq(r_lo, reversed prefix challenges) * eq(r_hi, .)
    (:2427-2466)."""

    def __init__(self, cycle_witnesses, r_spartan, gamma, d_rows=None):
        r = np.ascontiguousarray(r_spartan, dtype=np.uint64).reshape(-1, 4)
        n = r.shape[0]
        assert n >= 2
        own = None
        if d_rows is None:
            w = np.ascontiguousarray(cycle_witnesses, dtype=np.uint64).reshape(-1, NUM_R1CS_INPUTS, 4)
            assert w.shape[0] == (1 << n)  # the padded trace (the reference's own fold of an unpadded witness table drops odd tails)
            own = lib.DeviceBuffer.from_host(w)
            d_rows = own.ptr
        split = n // 2
        self.r_hi, self.r_lo = r[:split].copy(), r[split:].copy()
        ps, ss = 1 << (n - split), 1 << split
        self.gamma = np.ascontiguousarray(gamma, dtype=np.uint64).reshape(4).copy()
        g = fr_to_int(self.gamma)
        m = [0] * (NUM_R1CS_INPUTS + 1)
        m[_I["RdWriteValue"]], m[_I["Rs1Value"]], m[_I["Rs2Value"]] = 1, g, g * g % R_MOD
        N = 1 << n
        buf, ptrs = _witness_maps_dev(d_rows, N, N, [m] + [_column_map(c) for c in ("RdWriteValue", "Rs1Value", "Rs2Value")])
        q = _colsum_dev(ptrs[0], ss, ps, lib.fr_eq_table(self.r_hi).reshape(1, ss, 4))[0]
        self._rounds = RegistersClaimReductionRounds([lib.fr_eq_table(self.r_lo), q], self.gamma)
        self._wit = lib.ProductSumcheckSession.open_dev(ptrs[1:], N)
        lib.sync()
        buf.free()
        if own is not None:
            own.free()
        self.current_prefix_size, self.in_phase2, self.prefix_challenges = ps, False, []

    def computeRoundEvals(self, previous_claim):
        return self._rounds.computeRoundEvals(previous_claim)

    def bind(self, r_j):
        """:2388-2398"""
        r_j = np.ascontiguousarray(r_j, dtype=np.uint64).reshape(4)
        self._rounds.bind(r_j)
        if self.in_phase2:
            return
        transition = self.current_prefix_size == 2
        self._wit.bind(r_j)
        self.prefix_challenges.append(r_j.copy())
        self.current_prefix_size //= 2
        if transition:
            e = 1  # EqPolynomial(r_lo).evaluate(reversed prefix challenges)
            for a, b in zip((fr_to_int(x) for x in self.r_lo), (fr_to_int(x) for x in self.prefix_challenges[::-1])):
                e = e * ((a * b + (1 - a) * (1 - b)) % R_MOD) % R_MOD
            suffix = lib.fr_eq_table(self.r_hi)
            eq2 = lib.field_op(lib.FR, lib.OP_MUL, np.tile(fr_from_int(e), (suffix.shape[0], 1)), suffix)
            wit = [self._wit.read(j) for j in range(3)]
            self._wit.close()
            self._wit = None
            self._rounds.deinit()
            self._rounds = RegistersClaimReductionRounds([eq2] + wit, self.gamma, phase2=True)
            self.in_phase2 = True

    def finalClaims(self):
        return self._rounds.finalClaims()

    def deinit(self):
        if self._wit is not None:
            self._wit.close()
        self._rounds.deinit()


class Stage3Prover:
    """The round loop of Stage3Prover.generateStage3Proof (src/zkvm/spartan/stage3_prover.zig:113-760) over the three instances: input
    claims (:775-844), the combined cubic under the three batching coefficients, its compressed form (c0, c2, c3), and after a challenge
    every instance's claim from its own polynomial (evalsToCoeffs / evaluatePolyAtPoint, :846-926). The transcript stays the caller's:
    computeRoundPolynomial() -> compressed, bindChallenge(r_j)."""

    def __init__(self, cycle_witnesses, r_outer, r_product, shift_gamma_powers, instr_gamma, reg_gamma, input_claims, batching_coeffs, d_rows=None):
        """the witness matrix is uploaded once (or taken from HBM: d_rows) and read by all three instances"""
        r_outer = np.ascontiguousarray(r_outer, dtype=np.uint64).reshape(-1, 4)
        r_product = np.ascontiguousarray(r_product, dtype=np.uint64).reshape(-1, 4)
        n = r_outer.shape[0]
        N = 1 << n
        own = None
        if d_rows is None:
            w = np.ascontiguousarray(cycle_witnesses, dtype=np.uint64).reshape(-1, NUM_R1CS_INPUTS, 4)
            assert w.shape[0] == N
            own = lib.DeviceBuffer.from_host(w)
            d_rows = own.ptr
        self.shift = ShiftPrefixSuffixProver(None, r_outer, r_product, shift_gamma_powers, d_rows=d_rows)
        self.reg = RegistersPrefixSuffixProver(None, r_outer, reg_gamma, d_rows=d_rows)
        buf, ptrs = _witness_maps_dev(d_rows, N, N, [_column_map(c) for c in ("FlagLeftOperandIsRs1", "Rs1Value", "FlagLeftOperandIsPC", "UnexpandedPC",
                                                                                "FlagRightOperandIsRs2", "Rs2Value", "FlagRightOperandIsImm", "Imm")])
        d_eq = lib.DeviceBuffer(2 * N * 32)
        lib.fr_eq_table_dev(r_outer, d_eq.ptr)
        lib.fr_eq_table_dev(r_product, d_eq.ptr + N * 32)
        lib.sync()
        self.instr = InstructionInputProver(None, instr_gamma, d_tables=ptrs + [d_eq.ptr, d_eq.ptr + N * 32], n=N)
        lib.sync()
        for b in (buf, d_eq, own):
            if b is not None:
                b.free()
        self.claims = [fr_to_int(c) for c in np.ascontiguousarray(input_claims, dtype=np.uint64).reshape(3, 4)]  # shift, instruction input, registers
        self.coeffs = [fr_to_int(c) for c in np.ascontiguousarray(batching_coeffs, dtype=np.uint64).reshape(3, 4)]
        self.combined_claim = sum(c * k for c, k in zip(self.claims, self.coeffs)) % R_MOD

    @staticmethod
    def inputClaims(outer, product, shift_gamma_powers, instr_gamma, reg_gamma):
        """computeShiftInputClaim / computeInstructionInputClaim / computeRegistersInputClaim (:775-844); outer / product: opening claims by
        polynomial name (Montgomery limbs) at Stage 1's r_cycle and at the product sumcheck's -> (3, 4)"""
        o = {k: fr_to_int(v) for k, v in outer.items()}
        p = {k: fr_to_int(v) for k, v in product.items()}
        g = [fr_to_int(x) for x in np.ascontiguousarray(shift_gamma_powers, dtype=np.uint64).reshape(5, 4)]
        gi, gr = fr_to_int(instr_gamma), fr_to_int(reg_gamma)
        shift = o["NextUnexpandedPC"] + g[1] * o["NextPC"] + g[2] * o["NextIsVirtual"] + g[3] * o["NextIsFirstInSequence"] + g[4] * (1 - p["NextIsNoop"])
        instr = o["RightInstructionInput"] + gi * o["LeftInstructionInput"] + gi * gi * (p["RightInstructionInput"] + gi * p["LeftInstructionInput"])
        reg = o["RdWriteValue"] + gr * o["Rs1Value"] + gr * gr * o["Rs2Value"]
        return np.stack([fr_from_int(x % R_MOD) for x in (shift, instr, reg)])

    @staticmethod
    def evalsToCoeffs(ev):
        """:846-901 (degree 2 or 3), canonical integers"""
        if len(ev) == 3:
            p0, p1, p2 = ev
            c2 = (p2 - 2 * p1 + p0) * pow(2, -1, R_MOD) % R_MOD
            return [p0 % R_MOD, (p1 - p0 - c2) % R_MOD, c2]
        p0, p1, p2, p3 = ev
        d1, d2, d3 = p1 - p0, p2 - p1, p3 - p2
        c3 = ((d3 - d2) - (d2 - d1)) * pow(6, -1, R_MOD) % R_MOD
        c2 = ((d2 - d1) * pow(2, -1, R_MOD) - 3 * c3) % R_MOD
        return [p0 % R_MOD, (d1 - c2 - c3) % R_MOD, c2, c3]

    def computeRoundPolynomial(self):
        """:333-445 -> (3, 4): c0, c2, c3 of the combined cubic"""
        insts = (self.shift, self.instr, self.reg)
        self.round_evals = [[fr_to_int(x) for x in p.computeRoundEvals(fr_from_int(c))] for p, c in zip(insts, self.claims)]
        full = [e if len(e) == 4 else e + [(3 * e[2] - 3 * e[1] + e[0]) % R_MOD] for e in self.round_evals]  # a quadratic at 3 (:415-417)
        comb = [sum(full[k][i] * self.coeffs[k] for k in range(3)) % R_MOD for i in range(4)]
        self.combined_coeffs = self.evalsToCoeffs(comb)
        return np.stack([fr_from_int(self.combined_coeffs[i]) for i in (0, 2, 3)])

    def bindChallenge(self, r_j):
        """:458-490"""
        r = fr_to_int(r_j)
        at = lambda cs: sum(c * pow(r, i, R_MOD) for i, c in enumerate(cs)) % R_MOD
        self.combined_claim = at(self.combined_coeffs)
        self.claims = [at(self.evalsToCoeffs(e)) for e in self.round_evals]
        for p in (self.shift, self.instr, self.reg):
            p.bind(r_j)

    def deinit(self):
        self.shift.deinit()
        self.reg.deinit()
        self.instr.deinit()


class ExpandingTable:
    """ExpandingTable (src/zkvm/lasso/expanding_table.zig:27-190): after k binds the table IS the eq table of the k challenges, first
    challenge on the index's top bit, times the initial value — rebuilt by the device's eq-table kernel at every bind (:83-99 doubles it on
    the host); condense (:144-161) = an element-wise product and sums over runs of 2^(round - out_bits) entries (weighted column sums of
    the transposed products)."""

    def __init__(self, max_rounds, initial=None):
        self.max_rounds, self.round = max_rounds, 0
        self._initial = fr_from_int(1) if initial is None else np.ascontiguousarray(initial, dtype=np.uint64).reshape(4).copy()
        self._r = []
        self.values = self._initial.reshape(1, 4).copy()

    def size(self):
        return self.values.shape[0]

    def bind(self, r):
        assert self.round < self.max_rounds
        self._r.append(np.ascontiguousarray(r, dtype=np.uint64).reshape(4).copy())
        self.values = lib.fr_eq_table(np.stack(self._r), self._initial)
        self.round += 1

    def get(self, index):
        return self.values[index]

    def getAll(self):
        return self.values

    def sum(self):
        return fr_from_int(sum(fr_to_int(x) for x in self.values) % R_MOD) if self.values.shape[0] <= 4096 else \
            lib.fr_weighted_colsum(self.values, self.values.shape[0], 1, np.tile(fr_from_int(1), (1, self.values.shape[0], 1)))[0, 0]

    def condense(self, weights, out_bits):
        w = np.ascontiguousarray(weights, dtype=np.uint64).reshape(-1, 4)
        assert w.shape[0] == self.values.shape[0] and out_bits <= self.round
        out_size, chunk = 1 << out_bits, 1 << (self.round - out_bits)
        prod = lib.field_op(lib.FR, lib.OP_MUL, self.values, w).reshape(out_size, chunk, 4)
        t = np.ascontiguousarray(prod.transpose(1, 0, 2)).reshape(-1, 4)  # chunk rows of out_size columns: a column sum per output
        return lib.fr_weighted_colsum(t, chunk, out_size, np.tile(fr_from_int(1), (1, chunk, 1)))[0]


class SpartanOuterProver:
    """The standard rounds of SpartanOuterProver (src/zkvm/spartan/outer.zig:364-407) over its working_vals table: LowToHigh sums and
    folds on a LOW_PAIR device session (the UniSkip first round of this prover is StreamingOuterProver.computeFirstRoundPoly's)."""

    def __init__(self, working_vals):
        w = np.ascontiguousarray(working_vals, dtype=np.uint64).reshape(-1, 4)
        self.current_len = w.shape[0]
        self._s = lib.SumcheckSession.open(w, lib.SC_LOW_PAIR) if self.current_len else None
        self.challenges = []

    def computeStandardRoundPoly(self):
        """[p(0), p(1), 2 p(1) - p(0)] (:364-388); a single entry left: [it, 0, 0]"""
        z = np.zeros(4, dtype=np.uint64)
        if self.current_len <= 1:
            return np.stack([self._s.final() if self.current_len == 1 else z, z, z])
        p0, p1 = self._s.round_sums()
        return np.stack([p0, p1, _fr_sub(_fr_add(p1, p1), p0)])

    def bindChallenge(self, challenge):
        """:391-407"""
        self.challenges.append(np.ascontiguousarray(challenge, dtype=np.uint64).copy())
        if self.current_len <= 1:
            return
        self._s.bind(challenge)
        self.current_len //= 2

    def deinit(self):
        if self._s is not None:
            self._s.close()


class Phase1Prover:
    """Phase1Prover (src/zkvm/spartan/prefix_suffix.zig:35-147): up to six P / Q pairs in one product session; a round is one pass of pair-sum
    terms (two pairs per term), a bind folds every buffer."""

    def __init__(self):
        self._pending, self._s, self.challenges, self.current_size = [], None, [], 0

    def addPair(self, P, Q):
        P = np.ascontiguousarray(P, dtype=np.uint64).reshape(-1, 4)
        Q = np.ascontiguousarray(Q, dtype=np.uint64).reshape(-1, 4)
        assert P.shape == Q.shape and self.current_size in (0, P.shape[0]) and self._s is None and len(self._pending) < 12
        self.current_size = P.shape[0]
        self._pending += [P, Q]

    def shouldTransition(self):
        return self.current_size <= 2

    def _open(self):
        if self._s is None:
            self._s = lib.ProductSumcheckSession.open(self._pending)
            k = len(self._pending) // 2
            self._terms = [((4 * t, 4 * t + 1, 4 * t + 2, 4 * t + 3), (), None, True) for t in range(k // 2)]
            if k % 2:
                self._terms.append(((2 * k - 2, 2 * k - 1), (), None))
            self._s.set_points(0b0011)
            self._pending = None

    def computeRoundEvals(self):
        """[g(0), g(1)] (:95-112)"""
        self._open()
        ev = self._s.round_expr(self._terms)
        return np.stack([ev[0], ev[1]])

    def bind(self, r):
        """:114-132"""
        self._open()
        self.challenges.append(np.ascontiguousarray(r, dtype=np.uint64).copy())
        self._s.bind(r)
        self.current_size //= 2

    def pairs(self):
        self._open()
        t = [self._s.read(j) for j in range(self._s.tables())]
        return [(t[2 * i], t[2 * i + 1]) for i in range(len(t) // 2)]

    def deinit(self):
        if self._s is not None:
            self._s.close()


def initShiftQBuffers(unexpanded_pc, pc, is_virtual, is_first_in_sequence, is_noop, suffix_0_outer, suffix_1_outer, suffix_0_product,
                      suffix_1_product, gamma_powers, prefix_size):
    """initShiftQBuffers (src/zkvm/spartan/prefix_suffix.zig:149-232): the double loop over x_hi / x_lo is a weighted column sum of every input
    table (zg_fr_weighted_colsum, two suffixes per pass); the gamma combination and the (1 - noop) form are applied to the sqrt(T)-sized
    sums -> (Q_0_outer, Q_1_outer, Q_0_product, Q_1_product)"""
    g = [fr_to_int(x) for x in np.ascontiguousarray(gamma_powers, dtype=np.uint64).reshape(-1, 4)]
    so = np.stack([np.ascontiguousarray(suffix_0_outer, dtype=np.uint64).reshape(-1, 4), np.ascontiguousarray(suffix_1_outer, dtype=np.uint64).reshape(-1, 4)])
    sp = np.stack([np.ascontiguousarray(suffix_0_product, dtype=np.uint64).reshape(-1, 4), np.ascontiguousarray(suffix_1_product, dtype=np.uint64).reshape(-1, 4)])
    ss = so.shape[1]
    sums = [lib.fr_weighted_colsum(np.ascontiguousarray(t, dtype=np.uint64).reshape(-1, 4), ss, prefix_size, so) for t in (unexpanded_pc, pc, is_virtual, is_first_in_sequence)]
    noop = lib.fr_weighted_colsum(np.ascontiguousarray(is_noop, dtype=np.uint64).reshape(-1, 4), ss, prefix_size, sp)
    out = []
    for k in range(2):
        q = [[fr_to_int(x) for x in s_[k]] for s_ in sums]
        out.append(np.stack([fr_from_int((a + g[1] * b + g[2] * c + g[3] * d) % R_MOD) for a, b, c, d in zip(*q)]))
    for k in range(2):
        total = sum(fr_to_int(x) for x in sp[k]) % R_MOD  # sum over x_hi of the suffix: the "1" of (1 - noop)
        out.append(np.stack([fr_from_int(g[4] * (total - fr_to_int(x)) % R_MOD) for x in noop[k]]))
    return tuple(out)


class LassoPrefixPolynomial:
    """PrefixPolynomial (src/zkvm/lasso/prefix_suffix.zig:133-231): bind = the high-half fold (:175-196, zg_fr_bind_high), evaluate = the
    dense evaluation with the index's low bit on point[0] (:198-216, zg_fr_dense_evaluate)"""

    def __init__(self, evaluations, prefix_type=None):
        self.evaluations = np.ascontiguousarray(evaluations, dtype=np.uint64).reshape(-1, 4).copy()
        self.num_vars = self.evaluations.shape[0].bit_length() - 1
        self.prefix_type = prefix_type

    def get(self, index):
        return self.evaluations[index]

    def set(self, index, value):
        self.evaluations[index] = value

    def bind(self, challenge):
        assert self.num_vars > 0
        return LassoPrefixPolynomial(lib.fr_bind_high(self.evaluations, challenge), self.prefix_type)

    def evaluate(self, point):
        point = np.ascontiguousarray(point, dtype=np.uint64).reshape(-1, 4)
        assert point.shape[0] == self.num_vars
        return lib.fr_dense_evaluate(self.evaluations, point) if self.num_vars else self.evaluations[0].copy()


class JoltOuterProver:
    """JoltOuterProver's round loop (src/zkvm/spartan/jolt_outer_prover.zig:148-262) over the table f(cycle) = eq(tau, cycle) * Az * Bz it
    builds in init (:96-116, host work over the R1CS constraints): LowToHigh sums and folds on a LOW_PAIR device session."""

    def __init__(self, working_evals):
        w = np.ascontiguousarray(working_evals, dtype=np.uint64).reshape(-1, 4)
        self.current_len = w.shape[0]
        self._s = lib.SumcheckSession.open(w, lib.SC_LOW_PAIR)
        self.challenges = []
        self.current_claim = self._sum()  # :104-115: the sum of the table

    def _sum(self):
        if self.current_len >= 2:
            g0, g1 = self._s.round_sums()
            return _fr_add(g0, g1)
        return self._s.final()

    def computeRoundPoly(self):
        """[p(0), p(2)] with p(2) = 2 p(1) - p(0) (:152-176)"""
        if self.current_len <= 1:
            return np.stack([self.current_claim, np.zeros(4, dtype=np.uint64)])
        p0, p1 = self._s.round_sums()
        return np.stack([p0, _fr_sub(_fr_add(p1, p1), p0)])

    def computeCubicRoundPoly(self):
        """[p(0), p(2), p(3)] by linear extrapolation from p(0), p(1) (:181-226)"""
        if self.current_len <= 1:
            z = np.zeros(4, dtype=np.uint64)
            return np.stack([self.current_claim, z, z])
        p0, p1 = self._s.round_sums()
        a0, a1 = fr_to_int(p0), fr_to_int(p1)
        c1 = (a1 - a0) % R_MOD
        return np.stack([p0, fr_from_int((a0 + 2 * c1) % R_MOD), fr_from_int((a0 + 3 * c1) % R_MOD)])

    def bindChallenge(self, challenge):
        """:230-252: fold, the new claim is the sum of the folded table"""
        self.challenges.append(np.array(challenge, dtype=np.uint64))
        if self.current_len <= 1:
            return
        self._s.bind(challenge)
        self.current_len //= 2
        self.current_claim = self._sum()

    def getFinalEval(self):
        """:255-259"""
        if self.current_len == 0:
            return fr_from_int(0)
        return self._s.final() if self.current_len == 1 else self.current_claim

    def deinit(self):
        self._s.close()


class R1CSInputEvaluator:
    """R1CSInputEvaluator(F) (src/zkvm/r1cs/evaluation.zig:41-160): the MLE evaluations of all R1CS input columns at r_cycle from the
    cycle-major witness matrix — one eq table and one pass over the matrix on the device (zg_fr_rows_mle)."""

    @staticmethod
    def computeClaimedInputs(cycle_witnesses, r_cycle):
        """cycle_witnesses: (num_cycles, NUM_INPUTS, 4) = R1CSCycleInputs.values per cycle, or the shared device-resident
        CycleWitnessMatrix (no upload: zg_fr_rows_mle_dev over it); -> (NUM_INPUTS, 4)   (:55-122)"""
        if isinstance(cycle_witnesses, CycleWitnessMatrix):
            m = cycle_witnesses
            r = np.ascontiguousarray(r_cycle, dtype=np.uint64).reshape(-1, 4)
            if m.num_cycles == 0:
                return np.zeros((NUM_R1CS_INPUTS, 4), dtype=np.uint64)
            log_n = m.num_cycles.bit_length() - 1
            effective_len = min(r.shape[0], log_n)
            if effective_len == 0:
                return m.to_host()[0].copy()
            if effective_len < log_n:
                raise IndexError("computeClaimedInputs: r_cycle shorter than log2 of the cycle count")
            return lib.fr_rows_mle_dev(m.ptr, min(m.num_cycles, 1 << log_n), NUM_R1CS_INPUTS, r[:effective_len])
        w = np.ascontiguousarray(cycle_witnesses, dtype=np.uint64)
        assert w.ndim == 3 and w.shape[2] == 4
        r = np.ascontiguousarray(r_cycle, dtype=np.uint64).reshape(-1, 4)
        num_cycles, k = w.shape[0], w.shape[1]
        if num_cycles == 0:
            return np.zeros((k, 4), dtype=np.uint64)
        log_n = num_cycles.bit_length() - 1  # std.math.log2_int: floor
        padded_len = 1 << log_n
        effective_len = min(r.shape[0], log_n)
        if effective_len == 0:  # :75-83: no cycle variables — the first witness
            return w[0].copy()
        if effective_len < log_n:  # the reference indexes eq_evals[t] for t < padded_len: out of bounds here (a safety-checked panic)
            raise IndexError("computeClaimedInputs: r_cycle shorter than log2 of the cycle count")
        return lib.fr_rows_mle(w[:min(num_cycles, padded_len)], r[:effective_len])

    @staticmethod
    def computeClaimedInput(cycle_witnesses, r_cycle, input_index):
        """computeClaimedInput (:125-160): one column of the above"""
        return R1CSInputEvaluator.computeClaimedInputs(np.ascontiguousarray(cycle_witnesses, dtype=np.uint64)[:, input_index:input_index + 1], r_cycle)[0]


class SumcheckInstance:
    """SumcheckInstance(F) (src/zkvm/batched_sumcheck.zig:34-74): num_rounds, degree, input_claim and the three callbacks.
    compute_round_poly(round) -> [s(0), s(1), s(2), s(3)]; bind_challenge(challenge); cache_openings(r_sumcheck) optional."""

    def __init__(self, num_rounds, degree, input_claim, compute_round_poly, bind_challenge, cache_openings=None):
        self.num_rounds, self.degree = num_rounds, degree
        self.input_claim = np.ascontiguousarray(input_claim, dtype=np.uint64).copy()
        self.computeRoundPoly, self.bindChallenge = compute_round_poly, bind_challenge
        self.cacheOpenings = cache_openings or (lambda r: None)


class BatchedSumcheckProver:
    """BatchedSumcheckProver(F) (src/zkvm/batched_sumcheck.zig:77-262): several instances, possibly with fewer rounds than the
    longest, combined with transcript-sampled coefficients. The instances' round evaluations come from device sessions; the
    combination below is the reference's host algebra."""

    def __init__(self, inactive_scaling="proof_converter"):
        """inactive_scaling: the constant an instance contributes before its first round. "proof_converter" = coeff * claim *
        2^(start - round - 1), the loop `zolt prove` actually runs (src/zkvm/proof_converter.zig:3330-3343, Jolt's rule: twice the
        constant is the instance's share of the claim, so s(0) + s(1) = claim holds in every round — the captured Stage-2 run
        ends with "expected_batched == actual batched", logs/zolt.log:3583-3585). "batched_sumcheck_zig" = 2^(start - round), the
        formula of batched_sumcheck.zig:208-212 itself, which no caller in the reference reaches and which breaks that identity."""
        assert inactive_scaling in ("proof_converter", "batched_sumcheck_zig")
        self.inactive_scaling = inactive_scaling
        self.instances, self.batching_coeffs, self.challenges = [], [], []
        self.max_num_rounds = 0
        self.current_round = 0
        self.current_claim = fr_from_int(0)

    def addInstance(self, instance):  # :115-121
        self.instances.append(instance)
        self.max_num_rounds = max(self.max_num_rounds, instance.num_rounds)

    def setupBatching(self, transcript):
        """:127-186: absorb every input claim, sample one coefficient each (challengeScalarFull), claim = sum coeff * 2^k * claim"""
        for inst in self.instances:
            transcript.appendScalar(inst.input_claim)
        self.batching_coeffs = [transcript.challengeScalarFull() for _ in self.instances]
        self.current_claim = self.batchedClaim()

    def batchedClaim(self):
        acc = 0
        for inst, c in zip(self.instances, self.batching_coeffs):
            acc += fr_to_int(inst.input_claim) * pow(2, self.max_num_rounds - inst.num_rounds, R_MOD) * fr_to_int(c)
        return fr_from_int(acc % R_MOD)

    def combinedEvals(self):
        """:193-222 / proof_converter.zig:3026-3343: active instances contribute coeff * evals, the others a constant (see __init__)"""
        comb = [0, 0, 0, 0]
        for inst, c in zip(self.instances, self.batching_coeffs):
            ci = fr_to_int(c)
            start = self.max_num_rounds - inst.num_rounds
            if self.current_round >= start:
                ev = inst.computeRoundPoly(self.current_round - start)
                for j in range(4):
                    comb[j] = (comb[j] + fr_to_int(ev[j]) * ci) % R_MOD
            else:
                k = start - self.current_round - (1 if self.inactive_scaling == "proof_converter" else 0)
                w = fr_to_int(inst.input_claim) * pow(2, k, R_MOD) * ci % R_MOD
                for j in range(4):
                    comb[j] = (comb[j] + w) % R_MOD
        return np.stack([fr_from_int(v) for v in comb])

    def computeRoundPolynomial(self):
        """compressed [c0, c2, c3] (:224-226)"""
        return evalsToCompressed(self.combinedEvals())

    def bindChallenge(self, challenge):  # :229-241
        self.challenges.append(np.array(challenge, dtype=np.uint64))
        for inst in self.instances:
            if self.current_round >= self.max_num_rounds - inst.num_rounds:
                inst.bindChallenge(challenge)
        self.current_round += 1

    def updateClaim(self, round_evals, challenge):  # :244-247
        self.current_claim = cubicAtPoint(round_evals, challenge)

    def cacheOpenings(self):  # :250-258: each instance gets the suffix of challenges of its own rounds
        for inst in self.instances:
            inst.cacheOpenings(self.challenges[self.max_num_rounds - inst.num_rounds:])

    def getFinalClaim(self):
        return self.current_claim

    def numRounds(self):
        return self.max_num_rounds


def decompressRoundPoly(compressed, current_claim):
    """generateBatchedProof's recovery of [s(0), s(1), s(2), s(3)] from [c0, c2, c3] and the claim (:380-400):
    c1 = claim - 2 c0 - c2 - c3, then s(t) = c0 + c1 t + c2 t^2 + c3 t^3"""
    c0, c2, c3 = (fr_to_int(x) for x in compressed)
    c1 = (fr_to_int(current_claim) - 2 * c0 - c2 - c3) % R_MOD
    return np.stack([fr_from_int((c0 + c1 * t + c2 * t * t + c3 * t * t * t) % R_MOD) for t in range(4)])


def generateBatchedProof(prover, transcript):
    """generateBatchedProof (:306-430) with the Blake2b transcript: per round the compressed polynomial goes in between
    "UniPoly_begin" / "UniPoly_end", challengeScalar comes out. -> {round_polys (rounds,3,4), challenges, final_claim}"""
    polys, chals = [], []
    for _ in range(prover.numRounds()):
        comp = prover.computeRoundPolynomial()
        polys.append(comp)
        transcript.appendMessage(b"UniPoly_begin")
        for c in comp:
            transcript.appendScalar(c)
        transcript.appendMessage(b"UniPoly_end")
        challenge = transcript.challengeScalar()
        chals.append(challenge)
        full = decompressRoundPoly(comp, prover.current_claim)
        prover.updateClaim(full, challenge)
        prover.bindChallenge(challenge)
    z3, z1 = np.zeros((0, 3, 4), dtype=np.uint64), np.zeros((0, 4), dtype=np.uint64)
    return {"round_polys": np.stack(polys) if polys else z3, "challenges": np.stack(chals) if chals else z1, "final_claim": prover.getFinalClaim()}


class RamReadWriteCheckingProver:
    """RamReadWriteCheckingProver (src/zkvm/ram/read_write_checking.zig:160-1323): the three-phase sumcheck (phase1_num_rounds cycle
    variables, log_k address variables, the remaining cycle variables) over a SPARSE access list with three DENSE side tables, all of it
    behind one device session (zg_rwc_*): the library walks the list's integer fields on the host once per round (who pairs with whom,
    which checkpoint a lone entry meets — the reference's sequential loops), the coefficients and the tables live in HBM and every round
    is one kernel for the two sums and one for the bound list. What stays here is what the reference's struct keeps besides: the trace
    decoding of init, the GruenSplitEqPolynomial, the cubic, the claim. accesses: [(timestamp, address, is_write, value)] in trace order;
    initial_ram: {address: u64}."""

    def __init__(self, accesses, gamma, r_cycle, log_k, log_t, phase1_num_rounds, start_address, initial_claim, initial_ram=None, device_inc=False):
        self.gamma = np.ascontiguousarray(gamma, dtype=np.uint64).copy()
        self.r_cycle = np.ascontiguousarray(r_cycle, dtype=np.uint64).reshape(-1, 4).copy()
        self.log_k, self.log_t, self.phase1_num_rounds, self.start_address = log_k, log_t, phase1_num_rounds, start_address
        K, T = 1 << log_k, 1 << log_t
        inc = np.zeros((T, 4), dtype=np.uint64)
        val_init = np.zeros((K, 4), dtype=np.uint64)
        cur = {}
        for addr, val in (initial_ram or {}).items():  # :212-231, :253-267
            if addr >= start_address and (addr - start_address) // 8 < K:
                idx = (addr - start_address) // 8
                val_init[idx] = fr_from_int(val)
                cur[idx] = val
        ents = []
        for ts, address, is_write, value in accesses:  # :269-330
            if ts >= T or address < start_address or (address - start_address) // 8 >= K:
                continue
            idx = (address - start_address) // 8
            prev = cur.get(idx, 0)
            if is_write:
                inc[ts] = fr_from_int(value - prev)
                cur[idx] = value
            ents.append((ts, idx, prev if is_write else value, prev, value, bool(is_write)))
        ents.sort(key=lambda e: (e[0], e[1]))  # :333-340 (stable)
        col = lambda k, dt: np.array([e[k] for e in ents], dtype=dt)
        if device_inc:  # zg_rwc_open_writes: inc formed on the device from the write entries (refused when a cycle holds two writes)
            self._s = lib.RamRwSession.open_writes(log_k, log_t, col(0, np.uint32), col(1, np.uint32), col(2, np.uint64), col(3, np.uint64), col(4, np.uint64),
                                                   col(5, np.uint8), val_init, self.r_cycle)
        else:
            self._s = lib.RamRwSession.open(log_k, log_t, col(0, np.uint32), col(1, np.uint32), col(2, np.uint64), col(3, np.uint64), col(4, np.uint64),
                                            inc, val_init, self.r_cycle)
        self.eq_size = T
        self.gruen_eq = GruenSplitEqPolynomial(self.r_cycle)  # :354
        self.current_claim = fr_to_int(initial_claim)
        self.round = 0
        self.challenges = []
        self.last_q = None

    def numRounds(self):
        return self.log_k + self.log_t

    def isComplete(self):
        return self.round >= self.numRounds()

    def _in_cycle_phase(self):
        p1 = self.phase1_num_rounds
        return self.round < p1 or self.round >= p1 + self.log_k

    def computeRoundPolynomialCubic(self):
        """[s(0), s(1), s(2), s(3)] (:391-408)"""
        if self._in_cycle_phase():  # computePhase1Polynomial (:410-536) + Gruen's cubic
            d_out, n_out, d_in, n_in = self.gruen_eq.getWindowEqTablesDev(1)
            qc, qq = self._s.round_cycle(d_out, n_out, d_in, n_in, self.gamma)
            self.last_q = (fr_to_int(qc), fr_to_int(qq))
            return self.gruen_eq.computeCubicRoundPoly(qc, qq, fr_from_int(self.current_claim))
        addr_round = self.round - self.phase1_num_rounds  # computePhase2Polynomial (:538-769)
        s0, s2 = self._s.round_address(addr_round, self.challenges[self.phase1_num_rounds:self.phase1_num_rounds + addr_round], self.gamma)
        a, c = fr_to_int(s0), fr_to_int(s2)
        s1 = (self.current_claim - a) % R_MOD
        return np.stack([s0, fr_from_int(s1), s2, fr_from_int((3 * c - 3 * s1 + a) % R_MOD)])

    def bindChallenge(self, challenge):
        """bindChallenge (:902-970)"""
        ch = np.ascontiguousarray(challenge, dtype=np.uint64).copy()
        self.challenges.append(ch)
        p1 = self.phase1_num_rounds
        if self._in_cycle_phase() and self.eq_size > 1:
            self._s.bind_cycle(ch)  # eq_evals, inc and the entry list
            self.eq_size //= 2
            self.gruen_eq.bind(ch)
        if p1 <= self.round < p1 + self.log_k:
            self._s.bind_address(self.round - p1, ch)  # val_init and the entry list
        self.round += 1

    def updateClaim(self, evals, challenge):  # :1187-1204
        self.current_claim = fr_to_int(cubicAtPoint(np.asarray(evals, dtype=np.uint64).reshape(4, 4), challenge))

    def getOpeningClaims(self, r_sumcheck):
        """(ra_claim, val_claim, inc_claim) (:1210-1322)"""
        rs = np.asarray(r_sumcheck, dtype=np.uint64).reshape(-1, 4)
        log_k, log_t, p1 = self.log_k, self.log_t, self.phase1_num_rounds
        p2, p3 = p1 + log_k, log_t - p1
        r_address, r_cyc = np.zeros((log_k, 4), dtype=np.uint64), np.zeros((log_t, 4), dtype=np.uint64)
        for i in range(min(log_k, max(len(rs) - p1, 0))):
            r_address[log_k - 1 - i] = rs[p1 + i]
        for i in range(min(p1, len(rs))):
            if p3 + (p1 - 1 - i) < log_t:
                r_cyc[p3 + (p1 - 1 - i)] = rs[i]
        for i in range(min(p3, max(len(rs) - p2, 0))):
            r_cyc[p3 - 1 - i] = rs[p2 + i]
        return self._s.opening(r_address, r_cyc)

    # small accessors shared with the tests' checker
    def claim_element(self):
        return fr_from_int(self.current_claim)

    def cycle_scalars(self):
        return self._s.cycle_scalars()

    def entry_list(self):
        cyc, adr, ra, _, _, _ = self._s.read_entries()
        return [(int(c), int(a), fr_to_int(r)) for c, a, r in zip(cyc, adr, ra)]

    def entries_full(self):
        """[cycle, address, ra_coeff, val_coeff, prev_val, next_val] as integers, like the restatement's list"""
        cyc, adr, ra, val, prev, nxt = self._s.read_entries()
        return [[int(c), int(a), fr_to_int(r), fr_to_int(v), int(p), int(n)] for c, a, r, v, p, n in zip(cyc, adr, ra, val, prev, nxt)]

    def deinit(self):
        self._s.close()
        self.gruen_eq.deinit()


# the 19 uniform R1CS constraints (src/zkvm/r1cs/constraints.zig:248-531; the published Jolt R1CS): condition * (left - right) = 0, each side
# [(input index, coefficient)...] + constant over the 43 per-cycle inputs (R1CSInputIndex, :39-92)
NUM_R1CS_INPUTS = 43
_I = {n: i for i, n in enumerate((
    "LeftInstructionInput RightInstructionInput Product WriteLookupOutputToRD WritePCtoRD ShouldBranch PC UnexpandedPC Imm RamAddress Rs1Value "
    "Rs2Value RdWriteValue RamReadValue RamWriteValue LeftLookupOperand RightLookupOperand NextUnexpandedPC NextPC NextIsVirtual "
    "NextIsFirstInSequence LookupOutput ShouldJump FlagAddOperands FlagSubtractOperands FlagMultiplyOperands FlagLoad FlagStore FlagJump "
    "FlagWriteLookupOutputToRD FlagVirtualInstruction FlagAssert FlagDoNotUpdateUnexpandedPC FlagAdvice FlagIsCompressed FlagIsFirstInSequence "
    "FlagIsRdNotZero FlagBranch FlagIsNoop FlagLeftOperandIsRs1 FlagLeftOperandIsPC FlagRightOperandIsRs2 FlagRightOperandIsImm").split())}


def _lc(const=0, **terms):
    return ([(_I[k], v) for k, v in terms.items()], const)


UNIFORM_CONSTRAINTS = [
    (_lc(FlagLoad=1, FlagStore=1), _lc(RamAddress=1), _lc(Rs1Value=1, Imm=1)),
    (_lc(1, FlagLoad=-1, FlagStore=-1), _lc(RamAddress=1), _lc()),
    (_lc(FlagLoad=1), _lc(RamReadValue=1), _lc(RamWriteValue=1)),
    (_lc(FlagLoad=1), _lc(RamReadValue=1), _lc(RdWriteValue=1)),
    (_lc(FlagStore=1), _lc(Rs2Value=1), _lc(RamWriteValue=1)),
    (_lc(FlagAddOperands=1, FlagSubtractOperands=1, FlagMultiplyOperands=1), _lc(LeftLookupOperand=1), _lc()),
    (_lc(1, FlagAddOperands=-1, FlagSubtractOperands=-1, FlagMultiplyOperands=-1), _lc(LeftLookupOperand=1), _lc(LeftInstructionInput=1)),
    (_lc(FlagAddOperands=1), _lc(RightLookupOperand=1), _lc(LeftInstructionInput=1, RightInstructionInput=1)),
    (_lc(FlagSubtractOperands=1), _lc(RightLookupOperand=1), _lc(1 << 64, LeftInstructionInput=1, RightInstructionInput=-1)),
    (_lc(FlagMultiplyOperands=1), _lc(RightLookupOperand=1), _lc(Product=1)),
    (_lc(1, FlagAddOperands=-1, FlagSubtractOperands=-1, FlagMultiplyOperands=-1, FlagAdvice=-1), _lc(RightLookupOperand=1), _lc(RightInstructionInput=1)),
    (_lc(FlagAssert=1), _lc(LookupOutput=1), _lc(1)),
    (_lc(WriteLookupOutputToRD=1), _lc(RdWriteValue=1), _lc(LookupOutput=1)),
    (_lc(WritePCtoRD=1), _lc(RdWriteValue=1), _lc(4, UnexpandedPC=1, FlagIsCompressed=-2)),
    (_lc(ShouldJump=1), _lc(NextUnexpandedPC=1), _lc(LookupOutput=1)),
    (_lc(ShouldBranch=1), _lc(NextUnexpandedPC=1), _lc(UnexpandedPC=1, Imm=1)),
    (_lc(1, ShouldBranch=-1, FlagJump=-1), _lc(NextUnexpandedPC=1), _lc(4, UnexpandedPC=1, FlagDoNotUpdateUnexpandedPC=-4, FlagIsCompressed=-2)),
    (_lc(FlagVirtualInstruction=1), _lc(NextPC=1), _lc(1, PC=1)),
    (_lc(NextIsVirtual=1, NextIsFirstInSequence=-1), _lc(1), _lc(FlagDoNotUpdateUnexpandedPC=1)),
]
FIRST_GROUP_INDICES = (1, 2, 3, 4, 5, 6, 11, 14, 17, 18)  # constraints.zig:537-548
SECOND_GROUP_INDICES = (0, 7, 8, 9, 10, 12, 13, 15, 16)  # :553-563


def lagrangeEvals(r, size=10):
    """L_i(r) over the symmetric domain {-(size-1)/2 ..} (LagrangePoly.evals; computeLagrangeEvalsAtR0, streaming_outer.zig:1157-1213)"""
    rv, start = fr_to_int(r), -((size - 1) // 2)
    out = []
    for i in range(size):
        num = den = 1
        for j in range(size):
            if j != i:
                num = num * (rv - (start + j)) % R_MOD
                den = den * (i - j) % R_MOD
        out.append(fr_from_int(num * pow(den, R_MOD - 2, R_MOD) % R_MOD))
    return np.stack(out)


def lagrangeKernel(x, y, size=10):
    """LagrangePoly.lagrangeKernel (src/zkvm/r1cs/univariate_skip.zig:296-312): sum_i L_i(x) L_i(y)"""
    return fr_from_int(sum(fr_to_int(a) * fr_to_int(b) for a, b in zip(lagrangeEvals(x, size), lagrangeEvals(y, size))) % R_MOD)


def uniskipTargets(domain_size=10, degree=9):
    """uniskipTargets (src/zkvm/r1cs/univariate_skip.zig:188-225): -5, 6, -6, 7, ... for the outer sumcheck"""
    base_left = -((domain_size - 1) // 2)
    out, n, p = [], base_left - 1, base_left + domain_size
    while n >= -degree and p <= degree and len(out) < degree:
        out.append(n)
        if len(out) >= degree:
            break
        out.append(p)
        n, p = n - 1, p + 1
    while len(out) < degree and n >= -degree:
        out.append(n)
        n -= 1
    while len(out) < degree and p <= degree:
        out.append(p)
        p += 1
    return out


def shiftCoeffs(n, shift):
    """LagrangeHelper.shiftCoeffsI32 (univariate_skip.zig:398-448): p(shift) = sum_i alpha[i] p(i) for deg p < n"""
    def gb(t, k):
        if k == 0:
            return 1
        if t >= 0 and k > t:
            return 0
        tt, sign = (t, 1) if t >= 0 else (-t + k - 1, -1 if k & 1 else 1)
        num = den = 1
        for j in range(k):
            num, den = num * (tt - j), den * (j + 1)
        return sign * (num // den)
    return [(-1 if ((n - 1 - i) & 1) else 1) * gb(shift, i) * gb(shift - i - 1, n - 1 - i) for i in range(n)]


UNISKIP_TARGETS = uniskipTargets()
COEFFS_PER_J = [shiftCoeffs(10, t + 4) for t in UNISKIP_TARGETS]  # :469-476
PRODUCT_VIRTUAL_TARGETS = uniskipTargets(5, 4)  # -3, 3, -4, 4 (:56-59)
PRODUCT_VIRTUAL_COEFFS_PER_J = [shiftCoeffs(5, t + 2) for t in PRODUCT_VIRTUAL_TARGETS]  # :78-84


def interpolateIntDomain(vals, left):
    """coefficients (ascending) of the polynomial through (left + i, vals[i]) (lagrangeInterpolate, streaming_outer.zig:728-799)"""
    n = len(vals)
    coeffs = [0] * n
    for i, y in enumerate(vals):
        if y % R_MOD == 0:
            continue
        den, basis, deg = 1, [1] + [0] * (n - 1), 0
        for j in range(n):
            if j == i:
                continue
            den = den * (i - j) % R_MOD
            xj = left + j
            for k in range(deg + 1, 0, -1):
                basis[k] = (basis[k - 1] - xj * basis[k]) % R_MOD if k <= deg else basis[k - 1]
            basis[0] = (-xj * basis[0]) % R_MOD
            deg += 1
        scale = y * pow(den, R_MOD - 2, R_MOD) % R_MOD
        for k in range(n):
            coeffs[k] = (coeffs[k] + basis[k] * scale) % R_MOD
    return coeffs


class StreamingOuterProver:
    """StreamingOuterProver's remaining rounds (src/zkvm/spartan/streaming_outer.zig: init :120-212, bindFirstRoundChallenge :1135-1155,
    materializeLinearPhasePolynomials :258-372, computeRemainingRoundPoly :1215-1281, bindRemainingRoundChallenge :1681-1717, updateClaim
    :1723-1737). The cycle witnesses go to the device once; Az / Bz of both constraint groups are ONE launch over them (the Lagrange-weighted
    constraint sums are an affine map of a cycle's 43 inputs: zg_fr_rows_affine_dev), and live on as a two-table product session: a round
    is Gruen's (t'(0), t'(inf)) under the split-eq prefix tables (zg_psc_round_gruen) and a fold of both tables (zg_psc_bind). The
    split-eq scalar, the cubic and the claim are host algebra, as in the reference."""

    def __init__(self, cycle_witnesses, tau, lagrange_tau_r0=None):
        # a CycleWitnessMatrix (api.witness: built on the device from integer trace columns, shared with the other stages) or host rows
        self._owns_rows = not isinstance(cycle_witnesses, CycleWitnessMatrix)
        self._d_rows = CycleWitnessMatrix.from_witnesses(cycle_witnesses) if self._owns_rows else cycle_witnesses
        self.num_cycles = self._d_rows.num_cycles
        assert self.num_cycles > 0  # error.EmptyTrace (:147-149)
        self.padded_trace_len = 1
        while self.padded_trace_len < self.num_cycles:
            self.padded_trace_len *= 2
        self.num_cycle_vars = self.padded_trace_len.bit_length() - 1
        tau = np.ascontiguousarray(tau, dtype=np.uint64).reshape(-1, 4)
        assert tau.shape[0] == self.num_cycle_vars + 2
        self.tau_high = tau[-1].copy()
        self.full_tau = tau.copy()
        self.split_eq = GruenSplitEqPolynomial(tau[:-1], lagrange_tau_r0)
        self.current_claim = fr_from_int(0)
        self.current_round = 0
        self.challenges = []
        self.lagrange_evals_r0 = None
        self._s = None
        self.last_t = None

    def numRounds(self):
        return 1 + self.num_cycle_vars

    def computeFirstRoundPoly(self):
        """computeFirstRoundPoly (:523-597): the 28 coefficients of s1(Y) = L(tau_high, Y) t1(Y). t1 at the nine UniSkip targets is ONE
        launch over the resident witnesses: for target j and group g, Az(., Y_j) and Bz(., Y_j) are the COEFFS_PER_J[j]-weighted sums of the
        group's condition / left - right combinations — two affine maps of a cycle's inputs — and the launch sums their product under
        eq(tau_low, (cycle, group)) (zg_fr_rows_affine_prodsum_dev); the two interpolations and the product of the polynomials are host
        integer arithmetic (19 + 10 points)."""
        W = NUM_R1CS_INPUTS + 1
        m = [[0] * W for _ in range(36)]  # rows 2 p, 2 p + 1 = A_p, B_p for pair p = 2 j + g
        for j in range(9):
            for g, group in enumerate((FIRST_GROUP_INDICES, SECOND_GROUP_INDICES)):
                p = 2 * j + g
                for i, ci in enumerate(group):  # the second group uses the first nine of the ten coefficients (:631-657)
                    a = COEFFS_PER_J[j][i]
                    cond, left, right = UNIFORM_CONSTRAINTS[ci]
                    for row, lc, sign in ((2 * p, cond, 1), (2 * p + 1, left, 1), (2 * p + 1, right, -1)):
                        for idx, c in lc[0]:
                            m[row][idx] = (m[row][idx] + sign * a * c) % R_MOD
                        m[row][NUM_R1CS_INPUTS] = (m[row][NUM_R1CS_INPUTS] + sign * a * lc[1]) % R_MOD
        coeffs = np.stack([np.stack([fr_from_int(v) for v in row]) for row in m])
        tau_low = self.full_tau[:-1]
        d_w = lib.DeviceBuffer((1 << tau_low.shape[0]) * 32)
        lib.fr_eq_table_dev(tau_low, d_w.ptr)  # E_out[x_out] E_in[x_in] at index (x_out, x_in) = cycle * 2 + group (:541-566)
        out = lib.fr_rows_affine_prodsum_dev(self._d_rows.ptr, min(self.num_cycles, self.padded_trace_len), NUM_R1CS_INPUTS, coeffs, 18, d_w.ptr, 2)
        d_w.free()
        ext = [(fr_to_int(out[2 * j]) + fr_to_int(out[2 * j + 1])) % R_MOD for j in range(9)]
        self.last_extended_evals = ext
        t1 = [0] * 19
        for z, v in zip(UNISKIP_TARGETS, ext):
            t1[z + 9] = v
        t1_coeffs = interpolateIntDomain(t1, -9)
        lag_coeffs = interpolateIntDomain([fr_to_int(x) for x in lagrangeEvals(self.tau_high, 10)], -4)
        s1 = [0] * 28
        for i in range(10):
            for j in range(19):
                s1[i + j] = (s1[i + j] + lag_coeffs[i] * t1_coeffs[j]) % R_MOD
        return np.stack([fr_from_int(v) for v in s1])

    def bindFirstRoundChallenge(self, r0, uni_skip_claim):
        """r0 is not bound in split_eq: its weight is the initial scalar (:1135-1155)"""
        self.current_round = 1
        self.current_claim = np.ascontiguousarray(uni_skip_claim, dtype=np.uint64).copy()
        self.lagrange_evals_r0 = lagrangeEvals(r0, 10)

    def constraintMatrix(self):
        """(4, 44, 4): rows az(group 0), az(group 1), bz(group 0), bz(group 1) as affine maps of a cycle's inputs — sum_t L_t(r0) *
        condition_t and sum_t L_t(r0) * (left_t - right_t) over the group's constraints (:300-345), the constant in the last column"""
        wts = [fr_to_int(x) for x in self.lagrange_evals_r0]
        m = [[0] * (NUM_R1CS_INPUTS + 1) for _ in range(4)]
        for g, group in enumerate((FIRST_GROUP_INDICES, SECOND_GROUP_INDICES)):
            for t, ci in enumerate(group):
                cond, left, right = UNIFORM_CONSTRAINTS[ci]
                for row, lc, sign in ((g, cond, 1), (2 + g, left, 1), (2 + g, right, -1)):
                    for idx, c in lc[0]:
                        m[row][idx] = (m[row][idx] + sign * wts[t] * c) % R_MOD
                    m[row][NUM_R1CS_INPUTS] = (m[row][NUM_R1CS_INPUTS] + sign * wts[t] * lc[1]) % R_MOD
        return np.stack([np.stack([fr_from_int(v) for v in row]) for row in m])

    def materializeLinearPhasePolynomials(self):
        """Az[2 i + group], Bz[2 i + group] for every cycle i (zero past the trace), straight into a device session (:258-372)"""
        n2 = 2 * self.padded_trace_len
        d_az, d_bz = lib.DeviceBuffer(n2 * 32), lib.DeviceBuffer(n2 * 32)
        lib.fr_rows_affine_dev(self._d_rows.ptr, min(self.num_cycles, self.padded_trace_len), NUM_R1CS_INPUTS, self.constraintMatrix(), 2, 2,
                               self.padded_trace_len, [d_az.ptr, d_bz.ptr])
        self._s = lib.ProductSumcheckSession.open_dev([d_az.ptr, d_bz.ptr], n2)
        lib.sync()
        d_az.free()
        d_bz.free()

    def computeRemainingRoundPoly(self):
        """[s(0), s(1), s(2), s(3)] (:1215-1281): t'(0), t'(inf) of buildTPrimePoly / computeTEvals with window 1, then Gruen's cubic"""
        if self.current_round == 1 and self._s is None:
            self.materializeLinearPhasePolynomials()
        d_out, n_out, d_in, n_in = self.split_eq.getWindowEqTablesDev(1)
        t0, t_inf = self._s.round_gruen((0, 1), d_out, n_out, d_in, n_in)
        self.last_t = (t0, t_inf)
        return self.split_eq.computeCubicRoundPoly(t0, t_inf, self.current_claim)

    def bindRemainingRoundChallenge(self, r):
        """split_eq first, then Az / Bz low-to-high (:1681-1717)"""
        r = np.ascontiguousarray(r, dtype=np.uint64)
        self.challenges.append(r.copy())
        self.split_eq.bind(r)
        self._s.bind(r)
        self.current_round += 1

    def updateClaim(self, round_poly, challenge):
        self.current_claim = cubicAtPoint(round_poly, challenge)

    def getFinalEval(self):
        return self.current_claim

    def finalAzBz(self):
        """(Az, Bz) at the bound point once every variable is bound"""
        f = self._s.final()
        return f[0], f[1]

    def deinit(self):
        if self._s is not None:
            self._s.close()
        if self._owns_rows:
            self._d_rows.free()
        self.split_eq.deinit()


class Stage4GruenProver:
    """Stage4GruenProver (src/zkvm/spartan/stage4_gruen_prover.zig:65-1240), the RegistersReadWriteChecking sumcheck: five dense
    K = 128 x T tables (val, rd_wa, ra = gamma rs1_ra + gamma^2 rs2_ra, rs1_ra, rs2_ra), inc[T] and the eq structure over the cycles;
    LOG_K + log T rounds — phase1_num_rounds cycle variables in Gruen form, the seven register variables, the remaining cycle
    variables under the merged dense eq table. The tables are built ON THE DEVICE from the per-cycle trace columns and stay in HBM
    (zg_rrw_*): a round is one pass for the sums and one for the folds. The host keeps what the reference's host keeps: the
    GruenSplitEqPolynomial (its prefix tables are device buffers too), Gruen's cubic and the claim algebra.
    steps: [(instruction u32, rd_value u64, is_noop)] or the three columns as arrays; r_cycle in ROUND order (:283-288)."""
    LOG_K, K = 7, 128
    _RS1_OPS = (0x13, 0x03, 0x67, 0x1B, 0x33, 0x3B, 0x23, 0x63)  # opcodes that read rs1 (:205-218)
    _RS2_OPS = (0x33, 0x3B, 0x23, 0x63)  # ... rs2 (:220-231)
    _NO_RD_OPS = (0x23, 0x63)  # stores and branches write no register (:233-236)

    def __init__(self, steps, gamma, r_cycle, phase1_num_rounds, phase2_num_rounds, host_register_file=False):
        if isinstance(steps, tuple) and len(steps) == 3 and hasattr(steps[0], "__len__"):
            instr, rd_value, noop = (np.asarray(c) for c in steps)
        else:
            instr = np.array([s[0] for s in steps], dtype=np.uint32)
            rd_value = np.array([s[1] for s in steps], dtype=np.uint64)
            noop = np.array([bool(s[2]) for s in steps], dtype=bool)
        n = len(instr)
        T = 1
        while T < n:
            T *= 2
        self.T, self.log_T = T, T.bit_length() - 1
        r_cycle = np.ascontiguousarray(r_cycle, dtype=np.uint64).reshape(-1, 4)
        assert r_cycle.shape[0] == self.log_T and self.log_T >= 1
        assert 1 <= phase1_num_rounds <= self.log_T and phase2_num_rounds == self.LOG_K  # (the reference's configurations)
        self.num_rounds = self.LOG_K + self.log_T
        self.gamma = np.ascontiguousarray(gamma, dtype=np.uint64).copy()
        self.phase1_num_rounds, self.phase2_num_rounds = phase1_num_rounds, phase2_num_rounds
        if host_register_file:  # zg_rrw_open: the 32 x T register file and inc built here
            rs1, rs2, rd, reg_vals, inc = self.traceColumns(instr.astype(np.uint32), rd_value.astype(np.uint64), noop.astype(bool), T)
            self._s = lib.RegistersRwSession.open(self.log_T, rs1, rs2, rd, reg_vals, inc, self.gamma)
        else:  # zg_rrw_open_trace: both rebuilt on the device from the write column
            rs1, rs2, rd, written = self.traceWriteColumns(instr.astype(np.uint32), rd_value.astype(np.uint64), noop.astype(bool), T)
            self._s = lib.RegistersRwSession.open_trace(self.log_T, rs1, rs2, rd, written, self.gamma)
        self.gruen = GruenSplitEqPolynomial(r_cycle[::-1].copy())  # big-endian for the split-eq structure (:283-288)
        self.current_T, self.current_K = T, self.K
        self.last_q = None

    @classmethod
    def traceWriteColumns(cls, instr, rd_value, noop, T):
        """the columns zg_rrw_open_trace takes (:199-246): the register each cycle reads / writes (0xFF = none; rd only when it is used
        and non-zero) and the value written"""
        n = len(instr)
        opcode, live = instr & 0x7F, ~noop
        f_rd, f_rs1, f_rs2 = ((instr >> 7) & 31).astype(np.uint8), ((instr >> 15) & 31).astype(np.uint8), ((instr >> 20) & 31).astype(np.uint8)
        rs1 = np.full(T, 0xFF, dtype=np.uint8)
        rs2 = np.full(T, 0xFF, dtype=np.uint8)
        rd = np.full(T, 0xFF, dtype=np.uint8)
        written = np.zeros(T, dtype=np.uint64)
        rs1[:n] = np.where(live & np.isin(opcode, cls._RS1_OPS), f_rs1, 0xFF)
        rs2[:n] = np.where(live & np.isin(opcode, cls._RS2_OPS), f_rs2, 0xFF)
        writes = live & ~np.isin(opcode, cls._NO_RD_OPS) & (f_rd != 0)
        rd[:n] = np.where(writes, f_rd, 0xFF)
        written[:n] = np.where(writes, rd_value, 0)
        return rs1, rs2, rd, written

    @classmethod
    def traceColumns(cls, instr, rd_value, noop, T):
        """what initWithPhaseConfig reads of the trace (:183-258), as columns for zg_rrw_open: the register each cycle reads / writes
        (0xFF = none), the register file BEFORE every cycle (32 x T, padding cycles keep the last one) and
        inc = F.fromU64(post) - F.fromU64(pre) — the host-side form of what zg_rrw_open_trace rebuilds on the device."""
        n = len(instr)
        opcode, live = instr & 0x7F, ~noop
        f_rd = ((instr >> 7) & 31).astype(np.uint8)
        writes = live & ~np.isin(opcode, cls._NO_RD_OPS) & (f_rd != 0)
        rs1, rs2, rd, _ = cls.traceWriteColumns(instr, rd_value, noop, T)
        reg_vals = np.zeros((32, T), dtype=np.uint64)
        idx = np.arange(n)
        for k in range(1, 32):  # value of register k before cycle j = rd_value of its last write at a cycle < j
            last = np.maximum.accumulate(np.where(writes & (f_rd == k), idx, -1))
            prev = np.full(T, last[-1] if n else -1, dtype=np.int64)  # padding cycles: the final register file (:249-258)
            if n:
                prev[0] = -1
                prev[1:n] = last[:-1]
            reg_vals[k] = np.where(prev >= 0, rd_value[np.maximum(prev, 0)], 0)
        pre = np.zeros(T, dtype=np.uint64)
        post = np.zeros(T, dtype=np.uint64)
        w = np.nonzero(writes)[0]
        pre[w] = reg_vals[f_rd[w], w]
        post[w] = rd_value[w]
        def from_u64(u):  # F.fromU64
            l = np.zeros((T, 4), dtype=np.uint64)
            l[:, 0] = u
            return lib.field_op(lib.FR, lib.OP_TO_MONT, l)
        inc = lib.field_op(lib.FR, lib.OP_SUB, from_u64(post), from_u64(pre))
        return rs1, rs2, rd, reg_vals, inc

    def computeRoundEvals(self, rnd, current_claim):
        """computeRoundEvals (:1165-1190) -> (4, 4): p(0), p(1), p(2), p(3)"""
        p1, p2 = self.phase1_num_rounds, self.phase2_num_rounds
        claim = np.ascontiguousarray(current_claim, dtype=np.uint64)
        if rnd < p1:  # phase1ComputeMessage (:561-741)
            d_out, n_out, d_in, n_in = self.gruen.getWindowEqTablesDev(1)
            q0, qx = self._s.round_cycle_gruen(d_out, n_out, d_in, n_in)
            self.last_q = (q0, qx)
            return self.gruen.computeCubicRoundPoly(q0, qx, claim)
        if rnd < p1 + p2 or self.current_T == 1:  # phase2ComputeMessage (:764-852); phase 3 with a single cycle left (:955-1013)
            e0, e2 = self._s.round_address()
            a, b, c = fr_to_int(e0), fr_to_int(claim), fr_to_int(e2)
            e1 = (b - a) % R_MOD
            return np.stack([e0, fr_from_int(e1), e2, fr_from_int((a - 3 * e1 + 3 * c) % R_MOD)])  # the quadratic's p(3) (:841-850)
        e0, e2, e3 = self._s.round_cycle()  # phase3ComputeMessage (:854-953)
        return np.stack([e0, fr_from_int((fr_to_int(claim) - fr_to_int(e0)) % R_MOD), e2, e3])

    def bindChallenge(self, rnd, challenge):
        """bindChallenge / bindPolynomials (:1047-1163, 1192-1216)"""
        p1, p2 = self.phase1_num_rounds, self.phase2_num_rounds
        ch = np.ascontiguousarray(challenge, dtype=np.uint64)
        if rnd < p1 or rnd >= p1 + p2:
            self._s.bind_cycle(ch)
            self.current_T //= 2
            if rnd < p1:
                self.gruen.bind(ch)
                if rnd == p1 - 1:  # gruen_eq.merge (gruen_eq.zig:119-146): the dense table over the cycle variables still unbound
                    self._s.set_eq(self.gruen.getFullEqTable())
        else:
            self._s.bind_address(ch)
            self.current_K //= 2

    def getFinalClaims(self):
        """getFinalClaims (:1219-1236)"""
        f = self._s.final()
        return {"val_claim": f["val"], "rs1_ra_claim": f["rs1_ra"], "rs2_ra_claim": f["rs2_ra"], "rd_wa_claim": f["rd_wa"], "inc_claim": f["inc"]}

    def finalCheck(self):
        """(eq_scalar, combined, expected) as bindChallenge prints them after the last round (:1196-1210)"""
        f = {k: fr_to_int(v) for k, v in self._s.final().items()}
        comb = (f["ra"] * f["val"] + f["rd_wa"] * (f["val"] + f["inc"])) % R_MOD
        return fr_from_int(f["eq"]), fr_from_int(comb), fr_from_int(f["eq"] * comb % R_MOD)

    def deinit(self):
        self._s.close()
        self.gruen.deinit()


class Stage4Prover(Stage4GruenProver):
    """the original Stage4Prover (src/zkvm/spartan/stage4_prover.zig:74-865) on the same device session: the dense eq table over the cycles
    is set at the start (computeEqEvalsBE of the reversed r_cycle, :279-292), every cycle variable is bound first, then the seven register
    variables; all four evaluations of a round come from the tables (zg_rrw_round_cycle / zg_rrw_round_address with e1; the register rounds'
    p(3) = p(0) - 3 p(1) + 3 p(2), the polynomial being quadratic there), the full-coefficient round polynomial and prove()'s batched
    transcript loop are host code."""

    def __init__(self, steps, gamma, r_cycle, stage3_claims=None, batching_coeff=None):
        n = len(steps[0]) if isinstance(steps, tuple) and len(steps) == 3 and hasattr(steps[0], "__len__") else len(steps)
        log_t = max(n - 1, 0).bit_length()
        super().__init__(steps, gamma, r_cycle, max(log_t, 1), self.LOG_K)
        self._s.set_eq(self.gruen.getFullEqTable())  # eq(r_cycle_be, .) with r_cycle_be[0] <-> MSB
        self.stage3_claims = stage3_claims
        self.batching_coeff = fr_from_int(1) if batching_coeff is None else np.ascontiguousarray(batching_coeff, dtype=np.uint64).copy()

    def computeRoundEvals(self, rnd, current_claim=None):
        """computeRoundEvalsInternal (:601-723): p(0), p(1), p(2), p(3) from the tables"""
        if rnd < self.log_T:
            return np.stack(self._s.round_cycle(with_e1=True))
        e0, e1, e2 = self._s.round_address(with_e1=True)
        a, b, c = fr_to_int(e0), fr_to_int(e1), fr_to_int(e2)
        return np.stack([e0, e1, e2, fr_from_int((a - 3 * b + 3 * c) % R_MOD)])

    def computeRoundPolynomial(self, rnd, current_claim=None):
        """computeRoundPolynomial (:731-758) -> coefficients [c0, c1, c2, c3]"""
        e = [fr_to_int(x) for x in self.computeRoundEvals(rnd, current_claim)]
        c3 = (-e[0] + 3 * e[1] - 3 * e[2] + e[3]) * pow(6, R_MOD - 2, R_MOD) % R_MOD
        c2 = (2 * e[0] - 5 * e[1] + 4 * e[2] - e[3]) * pow(2, R_MOD - 2, R_MOD) % R_MOD
        return np.stack([fr_from_int(v) for v in (e[0], (e[1] - e[0] - c2 - c3) % R_MOD, c2, c3)])

    def bindChallenge(self, rnd, challenge):
        """bindPolynomials (:779-839)"""
        ch = np.ascontiguousarray(challenge, dtype=np.uint64)
        if rnd < self.log_T:
            self._s.bind_cycle(ch)
            self.current_T //= 2
        else:
            self._s.bind_address(ch)
            self.current_K //= 2

    def prove(self, transcript, input_claim=None):
        """prove (:395-567): the input claim from the Stage-3 claims when given (else input_claim, which the caller computed — the
        reference sums the full tables on the CPU, :569-599), batched compressed coefficients into the Blake2b transcript, the challenge,
        the batched claim"""
        b = fr_to_int(self.batching_coeff)
        if self.stage3_claims is not None:
            g = fr_to_int(self.gamma)
            rd, r1, r2 = (fr_to_int(x) for x in self.stage3_claims)
            unbatched = (rd + g * r1 + g * g * r2) % R_MOD
        else:
            unbatched = fr_to_int(input_claim)
        claim = unbatched * b % R_MOD
        polys, chals = [], []
        for rnd in range(self.num_rounds):
            c = [fr_to_int(x) for x in self.computeRoundPolynomial(rnd, fr_from_int(claim))]
            for i in (0, 2, 3):
                transcript.appendScalar(fr_from_int(c[i] * b % R_MOD))
            ch = transcript.challengeScalar()
            chals.append(ch)
            x = fr_to_int(ch)
            claim = (c[0] + x * (c[1] + x * (c[2] + x * c[3]))) % R_MOD * b % R_MOD
            self.bindChallenge(rnd, ch)
            polys.append(np.stack([fr_from_int(v * b % R_MOD) for v in c]))
        out = self.getFinalClaims()
        out.update({"round_polys": np.stack(polys), "challenges": np.stack(chals), "final_claim": fr_from_int(claim)})
        return out


class LassoProver:
    """LassoProver's sumcheck over eq_evals (src/zkvm/lasso/prover.zig:80-467) on ONE device session: the padded eq_evals array is
    built on the device (SplitEqPolynomial.getEq, src/zkvm/lasso/split_eq.zig:113-168 = the eq table with each half's variables
    reversed), the log_K address rounds split / scale it by index bits, the log_T cycle rounds are HIGH_HALF sums and folds.
    The prefix-suffix structures the reference binds alongside (:402-404) do not enter the round polynomials and are not mirrored."""

    def __init__(self, lookup_indices_u128, log_T, log_K, r_reduction):
        idx = np.ascontiguousarray(lookup_indices_u128, dtype=np.uint64).reshape(-1, 2)
        w = np.ascontiguousarray(r_reduction, dtype=np.uint64).reshape(-1, 4)
        assert w.shape[0] == log_T  # SplitEqPolynomial.init: w.len == num_outer + num_inner
        self.log_T, self.log_K = log_T, log_K
        self.num_cycles = idx.shape[0]
        padded = 1 << log_T
        assert self.num_cycles <= padded
        outer = log_T // 2  # :129-131
        # E_out / E_in are built LSB-first (w[i] <-> bit i of the half's index): MSB-first order = each half reversed
        point = np.concatenate([w[:outer][::-1], w[outer:][::-1]]) if log_T else w
        buf = lib.DeviceBuffer(padded * 32)
        lib.fr_eq_table_dev(point, buf.ptr)
        if self.num_cycles < padded:  # :160-164: cycles without lookups are zero
            tail = np.zeros((padded - self.num_cycles, 4), dtype=np.uint64)
            lib._chk(lib._lib.zg_memcpy_h2d(lib._d(buf.ptr + self.num_cycles * 32), lib._h(tail), tail.nbytes), "zg_memcpy_h2d")
        self._s = lib.SumcheckSession.open_dev(buf.ptr, padded, lib.SC_HIGH_HALF)
        lib.sync()
        buf.free()
        self._idx = lib.DeviceBuffer.from_host(idx) if self.num_cycles else lib.DeviceBuffer(16)
        self.eq_evals_len = padded
        self.round = 0
        self.challenges = []
        self.current_claim = self._total()  # :166-171

    def _total(self):
        if len(self._s) >= 2:
            g0, g1 = self._s.round_sums()
            return _fr_add(g0, g1)
        return self._s.final()

    def computeInitialClaim(self):
        return self.current_claim.copy()

    def isAddressPhase(self):
        return self.round < self.log_K

    def isComplete(self):
        return self.round >= self.log_K + self.log_T

    def computeRoundPolynomial(self):
        """-> coeffs [c0, c1, c2] = [sum_0, sum_1 - sum_0, 0] (:262-345)"""
        zero = np.zeros(4, dtype=np.uint64)
        if self.isAddressPhase():
            s0, s1 = self._s.bit_round(self._idx.ptr, self.num_cycles, self.round)
        else:
            if self.eq_evals_len <= 1:  # :325-333
                return np.stack([self._s.final(), zero, zero])
            s0, s1 = self._s.round_sums()
        return np.stack([s0, _fr_sub(s1, s0), zero])

    def receiveChallenge(self, challenge):
        """:352-453 — address phase: scale by the index bit, claim = sum; cycle phase: high-half fold, claim = sum of the folded array"""
        challenge = np.ascontiguousarray(challenge, dtype=np.uint64)
        self.challenges.append(challenge.copy())
        if self.isAddressPhase():
            self.current_claim = self._s.bit_bind(self._idx.ptr, self.num_cycles, self.round, challenge)
        elif self.eq_evals_len > 1:
            self._s.bind(challenge)
            self.eq_evals_len //= 2
            self.current_claim = self._total()
        self.round += 1

    def getFinalEval(self):
        """getFinalEval (:458-462) = expanding_v.get(0) = prod over the address challenges of (1 - r) (expanding_table.zig:83-99)"""
        assert self.isComplete()
        acc = 1
        for c in self.challenges[:self.log_K]:
            acc = acc * (1 - fr_to_int(c)) % R_MOD
        return fr_from_int(acc)

    def getChallenges(self):
        return np.stack(self.challenges) if self.challenges else np.zeros((0, 4), dtype=np.uint64)

    def eq_evals(self):
        """the live prefix of eq_evals, read back (tests)"""
        return self._s.read()

    def deinit(self):
        self._s.close()
        self._idx.free()


def lassoDeriveChallenge(coeffs, round_index):
    """deriveChallenge (src/zkvm/lasso/prover.zig:533-551): a 64-bit mix of the round index and the coefficients' Montgomery limbs"""
    M = (1 << 64) - 1
    h = 0x9E3779B97F4A7C15 ^ round_index
    h = h * 0xFF51AFD7ED558CCD & M
    for c in np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4):
        for limb in c:
            h ^= int(limb)
            h = h * 0xC4CEB9FE1A85EC53 & M
    h ^= h >> 33
    return fr_from_int(h)


def runLassoProver(lookup_indices_u128, log_T, log_K, r_reduction):
    """runLassoProver (:495-530) -> {round_polys (rounds,3,4), final_eval, challenges}"""
    p = LassoProver(lookup_indices_u128, log_T, log_K, r_reduction)
    polys = []
    rnd = 0
    while not p.isComplete():
        polys.append(p.computeRoundPolynomial())
        p.receiveChallenge(lassoDeriveChallenge(polys[-1], rnd))
        rnd += 1
    out = {"round_polys": np.stack(polys) if polys else np.zeros((0, 3, 4), dtype=np.uint64), "final_eval": p.getFinalEval(),
           "challenges": p.getChallenges()}
    p.deinit()
    return out


__all__ = [_k for _k in dir() if not _k.startswith("__")]  # underscore helpers are shared between the parts too
