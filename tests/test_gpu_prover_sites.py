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
    assert prover.isComplete() and np.array_equal(prover.getFinalClaim(), cur[0])
    if log_k <= 10:  # a true claim stays consistent: the final claim equals ra(r) * unmap(r)
        unmap_r = (start + 8 * sum(U.fr_to_int(b) << j for j, b in enumerate(bound))) % api.R_MOD
        assert U.fr_to_int(wclaim) == U.fr_to_int(cur[0]) * unmap_r % api.R_MOD
    prover.deinit()


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
