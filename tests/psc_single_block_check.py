"""Run by tests/test_gpu_psc_grid.py in a child process with ZG_PSC_BLOCKS=1 (the grid cap is read once per process): one workgroup
walks the whole table, so every thread owns 128 pairs of a 2^16-entry table in round 0 and 64 in the fused fold of round 1 — the
regime in which the lazy limb sums of psc.hip (ChainAcc4: a carry pass every four additions, a reduction every 64) and the Gruen
kernel's Acc29 are flushed INSIDE the loop, which the default grid only reaches at 2^25 entries. Prover loops against the oracle."""
import sys

import numpy as np


def main():
    from oracle import binding as ob
    from tests import util as U
    from zolt_amd import api, lib
    lib.init()
    v, rounds = 16, 4
    n = 1 << v
    rnd = lambda seed, k: ob.f_to_mont(ob.FR, U.random_raw256(seed, k))
    ch = rnd(1, rounds)
    ch[1, :2] = 0  # one narrow challenge among them
    ch[1, 3] &= np.uint64((1 << 61) - 1)
    claim = rnd(2, 1)[0]

    tabs = [rnd(10 + j, n) for j in range(5)]
    g, o = api.OutputSumcheckProver(*tabs, claim), ob.OutputSumcheckProver(*tabs, claim)
    for k in range(rounds):
        assert np.array_equal(g.roundEvals(), o.roundEvals()), ("output", k)
        g.bindChallenge(ch[k]); o.bindChallenge(ch[k])
    g.deinit()

    inc, wa, lt = (rnd(20 + j, n) for j in range(3))
    g, o = api.ValEvaluationProver(inc, wa, lt, claim), ob.ValEvaluationProver(inc, wa, lt, claim)
    for k in range(rounds):
        rp, wrp = g.computeRoundPolynomial(), o.computeRoundPolynomial()
        assert np.array_equal(rp, wrp), ("val_evaluation", k)
        g.bindChallengeWithPoly(ch[k], rp); o.bindChallengeWithPoly(ch[k], wrp)
    g.deinit()

    tabs = [rnd(30 + j, n) for j in range(10)]
    gamma = rnd(3, 1)[0]
    g = api.InstructionInputProver(tabs, gamma)
    cur, c = [t.copy() for t in tabs], claim
    for k in range(rounds):
        got, want = g.computeRoundEvals(c), ob.instruction_input_round(cur, gamma, c)
        assert np.array_equal(got, want), ("instruction_input", k)
        g.bind(ch[k])
        cur = [ob.fr_bind_low(t, ch[k]) for t in cur]
        c = ob.raf_update_claim(want, ch[k])
    g.deinit()

    tabs = [rnd(50 + j, n) for j in range(4)]
    g = api.InstructionLookupsClaimReductionProver(*tabs, gamma, claim)
    o = ob.InstructionLookupsClaimReduction(*tabs, gamma, claim)
    for k in range(rounds):
        ev, wev = g.computeRoundPolynomialCubic(), o.computeRoundPolynomialCubic()
        assert np.array_equal(ev, wev), ("instruction_lookups", k)
        g.bindChallenge(ch[k]); o.bindChallenge(ch[k])
        g.updateClaim(ev, ch[k]); o.updateClaim(wev, ch[k])
    g.deinit()

    left, right = rnd(60, n), rnd(61, n)
    tau = rnd(62, v)
    lk = rnd(63, 1)[0]
    g, o = api.ProductVirtualRemainderProver(left, right, tau, lk, claim), ob.ProductRemainderProver(left, right, tau, lk, claim)
    for k in range(rounds):
        ev, wev = g.roundEvals(), o.roundEvals()
        assert np.array_equal(ev, wev), ("product_remainder", k)
        g.bindChallenge(ch[k]); o.bindChallenge(ch[k])
        g.updateClaim(ev, ch[k]); o.updateClaim(wev, ch[k])
    g.deinit()
    print("single-block grid ok")


if __name__ == "__main__":
    sys.exit(main())
