"""Stage4GruenProver (RegistersReadWriteChecking, src/zkvm/spartan/stage4_gruen_prover.zig) with its five K x T tables in HBM
(zg_rrw_*): the device mirror against the reference's captured run and against the CPU restatement (oracle/) on seeded traces."""
import numpy as np
import pytest

from oracle import binding as ob
from tests.test_transcript_host import check_stage4_against_the_captured_run, stage4_inputs_of_the_captured_run

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from zolt_amd import api, lib
    lib.init(0)
    return api


def test_captured_run_on_the_device(api, golden_dir):
    """the reference's own Stage-4 run (K = 128 x T = 256, phases 4 / 7 / 4, its 15 challenges): the device mirror reproduces the
    printed round-0 evaluations, merged_eq[0], the combined value, their product and the final claim — the same check the oracle passes"""
    fx, gr, steps, gamma, r_cycle = stage4_inputs_of_the_captured_run(golden_dir, api.fr_from_int)
    p = api.Stage4GruenProver(steps, gamma, r_cycle, fx["phase1_num_rounds"], fx["phase2_num_rounds"])
    assert (p.T, p.K) == (fx["T"], fx["K"])
    check_stage4_against_the_captured_run(p, fx, gr, api.fr_from_int, api.fr_to_int)
    p.deinit()


_OPS = (0x13, 0x03, 0x67, 0x1B, 0x33, 0x3B, 0x23, 0x63, 0x37, 0x6F, 0x17)


def seeded_steps(seed, n, noop_share=0.1):
    """a synthetic trace: random opcodes of every class the prover distinguishes, random registers, random rd values, some no-ops"""
    rng = np.random.default_rng(seed)
    op = rng.choice(_OPS, size=n).astype(np.uint32)
    rd, rs1, rs2 = (rng.integers(0, 32, size=n).astype(np.uint32) for _ in range(3))
    instr = op | (rd << 7) | (rs1 << 15) | (rs2 << 20) | (rng.integers(0, 128, size=n).astype(np.uint32) << 25)
    val = rng.integers(0, 1 << 63, size=n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=n, dtype=np.uint64)
    noop = rng.random(n) < noop_share
    return [(int(i), int(v), bool(z)) for i, v, z in zip(instr, val, noop)]


@pytest.mark.parametrize("log_t,n_steps,p1", [(1, 2, 1), (3, 5, 1), (4, 16, 4), (6, 50, 3), (8, 256, 4), (10, 1000, 5)])
def test_rounds_against_the_restatement(api, log_t, n_steps, p1):
    """every round's four evaluations, the claim chain and the final claims, bit for bit; trace lengths that need padding, a phase-1
    length from a single round to all the cycle variables (no phase 3 then)"""
    steps = seeded_steps(100 + log_t, n_steps)
    rng = np.random.default_rng(log_t)
    r = ob.f_to_mont(ob.FR, rng.integers(0, 1 << 63, size=(log_t + 1 + 7 + log_t, 4), dtype=np.uint64))
    gamma, r_cycle, chals = r[0], r[1:1 + log_t], r[1 + log_t:]
    o = ob.Stage4GruenProver(steps, gamma, r_cycle, p1, 7)
    d = api.Stage4GruenProver(steps, gamma, r_cycle, p1, 7)
    claim = o.computeInputClaim()
    for k in range(7 + log_t):
        eo, ed = o.computeRoundEvals(k, claim), d.computeRoundEvals(k, claim)
        assert np.array_equal(eo, ed), k
        assert np.array_equal(ob.f_add(ob.FR, eo[0:1], eo[1:2])[0], claim), k
        claim = ob.raf_update_claim(eo, chals[k])
        o.bindChallenge(k, chals[k])
        d.bindChallenge(k, chals[k])
        assert (d.current_T, d.current_K) == (o.current_T, o.current_K)
    fo, fd = o.getFinalClaims(), d.getFinalClaims()
    for name in fo:
        assert np.array_equal(fo[name], fd[name]), name
    for a, b in zip(o.finalCheck(), d.finalCheck()):
        assert np.array_equal(a, b)
    assert np.array_equal(d.finalCheck()[2], claim)  # the last claim is eq * (ra val + wa (val + inc)) at the bound point
    d.deinit()


@pytest.mark.parametrize("log_t,n_steps,noop_share", [(1, 2, 0.0), (5, 31, 0.1), (6, 64, 0.0), (7, 100, 0.5), (12, 4000, 0.1), (13, 8192, 0.985), (14, 9000, 0.0)])
def test_register_file_rebuilt_on_the_device(api, log_t, n_steps, noop_share):
    """zg_rrw_open_trace (write column only; register file and inc rebuilt by the device's last-write scan) against zg_rrw_open (both
    built on the host by traceColumns): every round identical — traces shorter than a 64-cycle chunk, padded ones, and one with so few
    writes that a register's value is carried over dozens of chunks"""
    steps = seeded_steps(900 + log_t, n_steps, noop_share)
    rng = np.random.default_rng(50 + log_t)
    r = ob.f_to_mont(ob.FR, rng.integers(0, 1 << 63, size=(log_t + 1 + 7 + log_t, 4), dtype=np.uint64))
    gamma, r_cycle, chals = r[0], r[1:1 + log_t], r[1 + log_t:]
    p1 = max(1, log_t // 2)
    h = api.Stage4GruenProver(steps, gamma, r_cycle, p1, 7, host_register_file=True)
    d = api.Stage4GruenProver(steps, gamma, r_cycle, p1, 7)
    claim = api.fr_from_int(0)
    for k in range(7 + log_t):
        eh, ed = h.computeRoundEvals(k, claim), d.computeRoundEvals(k, claim)
        assert np.array_equal(eh, ed), k
        claim = ob.raf_update_claim(eh, chals[k])
        h.bindChallenge(k, chals[k])
        d.bindChallenge(k, chals[k])
    fh, fd = h.getFinalClaims(), d.getFinalClaims()
    for name in fh:
        assert np.array_equal(fh[name], fd[name]), name
    h.deinit()
    d.deinit()


def test_trace_columns_match_the_sequential_register_file(api):
    """traceColumns' vectorised register file against the reference's sequential loop (restated here): value before every cycle"""
    steps = seeded_steps(7, 300)
    instr = np.array([s[0] for s in steps], dtype=np.uint32)
    val = np.array([s[1] for s in steps], dtype=np.uint64)
    noop = np.array([s[2] for s in steps], dtype=bool)
    rs1, rs2, rd, reg_vals, inc = api.Stage4GruenProver.traceColumns(instr, val, noop, 512)
    regs = [0] * 32
    for j, (w, v, z) in enumerate(steps):
        assert [int(x) for x in reg_vals[:, j]] == regs
        op, r = w & 0x7F, (w >> 7) & 31
        wr = not z and op not in (0x23, 0x63) and r != 0
        assert rd[j] == (r if wr else 0xFF)
        if wr:
            assert api.fr_to_int(inc[j]) == (v - regs[r]) % api.R_MOD
            regs[r] = v
        else:
            assert not inc[j].any()
    for j in range(300, 512):
        assert [int(x) for x in reg_vals[:, j]] == regs and rd[j] == rs1[j] == rs2[j] == 0xFF


@pytest.mark.parametrize("log_t", [16])
def test_full_size_claim_chain(api, log_t):
    """2^16 cycles (five 256 MiB tables): too long for the restatement's numpy loops, so the size-independent property — every round's
    p(0) + p(1) is the running claim, starting from the input claim computed independently from the trace columns, and the last claim
    is eq * combined at the bound point."""
    n = (1 << log_t) - 37
    rng = np.random.default_rng(5)
    op = rng.choice(_OPS, size=n).astype(np.uint32)
    rdf, rs1f, rs2f = (rng.integers(0, 32, size=n).astype(np.uint32) for _ in range(3))
    instr = op | (rdf << 7) | (rs1f << 15) | (rs2f << 20)
    val = rng.integers(0, 1 << 63, size=n, dtype=np.uint64)
    noop = rng.random(n) < 0.05
    r = ob.f_to_mont(ob.FR, rng.integers(0, 1 << 63, size=(2 * log_t + 8, 4), dtype=np.uint64))
    gamma, r_cycle, chals = r[0], r[1:1 + log_t], r[1 + log_t:]
    p = api.Stage4GruenProver((instr, val, noop), gamma, r_cycle, log_t // 2, 7)
    # input claim from the columns: sum_j eq(r_cycle, j) [gamma val[rs1] + gamma^2 val[rs2] + (val[rd] + inc)] over the cycles that touch a register
    T = 1 << log_t
    rs1, rs2, rd, reg_vals, inc = p.traceColumns(instr, val, noop, T)
    eq = ob.fr_eq_table(r_cycle[::-1].copy())
    j = np.arange(T)
    g = gamma.reshape(1, 4)
    g2 = ob.f_mul(ob.FR, g, g)
    def fv(col):
        u = np.where(col != 0xFF, reg_vals[np.minimum(col, 31), j], 0).astype(np.uint64)
        return ob.f_from_u64(ob.FR, u)
    t1 = ob.f_mul(ob.FR, fv(rs1), np.repeat(g, T, axis=0))
    t2 = ob.f_mul(ob.FR, fv(rs2), np.repeat(g2, T, axis=0))
    t3 = ob.f_add(ob.FR, fv(rd), inc)  # inc is zero where nothing is written, and val[rd] = 0 there too
    per_cycle = ob.f_add(ob.FR, ob.f_add(ob.FR, t1, t2), t3)
    claim = ob._fsum(ob.f_mul(ob.FR, per_cycle, eq))
    for k in range(7 + log_t):
        ev = p.computeRoundEvals(k, claim)
        assert np.array_equal(ob.f_add(ob.FR, ev[0:1], ev[1:2])[0], claim), k
        claim = ob.raf_update_claim(ev, chals[k])
        p.bindChallenge(k, chals[k])
    assert np.array_equal(p.finalCheck()[2], claim)
    p.deinit()


@pytest.mark.parametrize("log_t,n_steps", [(1, 2), (4, 11), (8, 256), (10, 1000)])
def test_original_stage4_prover_against_the_restatement(api, log_t, n_steps):
    """Stage4Prover (stage4_prover.zig): cycle variables first under the dense eq table, all four evaluations from the tables — every
    round's evaluations and coefficient form, the final claims, and prove() through two Blake2b transcripts"""
    steps = seeded_steps(500 + log_t, n_steps)
    rng = np.random.default_rng(40 + log_t)
    r = ob.f_to_mont(ob.FR, rng.integers(0, 1 << 63, size=(2 * log_t + 12, 4), dtype=np.uint64))
    gamma, r_cycle, chals, coeff = r[0], r[1:1 + log_t], r[1 + log_t:8 + 2 * log_t], r[8 + 2 * log_t]
    o, d = ob.Stage4Prover(steps, gamma, r_cycle), api.Stage4Prover(steps, gamma, r_cycle)
    claim = o.computeInputClaim()
    for k in range(7 + log_t):
        eo, ed = o.computeRoundEvals(k, claim), d.computeRoundEvals(k, claim)
        assert np.array_equal(eo, ed), k
        assert np.array_equal(ob.f_add(ob.FR, eo[0:1], eo[1:2])[0], claim), k
        assert np.array_equal(o.computeRoundPolynomial(k), d.computeRoundPolynomial(k)), k
        claim = ob.raf_update_claim(eo, chals[k])
        o.bindChallenge(k, chals[k])
        d.bindChallenge(k, chals[k])
    fo, fd = o.getFinalClaims(), d.getFinalClaims()
    assert all(np.array_equal(fo[n], fd[n]) for n in fo)
    assert all(np.array_equal(a, b) for a, b in zip(o.finalCheck(), d.finalCheck())) and np.array_equal(d.finalCheck()[2], claim)
    d.deinit()
    # prove(): batching coefficient, Stage-3 claims absent -> the computed input claim
    o, d = ob.Stage4Prover(steps, gamma, r_cycle, None, coeff), api.Stage4Prover(steps, gamma, r_cycle, None, coeff)
    ta, tb = api.Blake2bTranscript(b"Jolt"), api.Blake2bTranscript(b"Jolt")
    want = o.prove(ta)
    got = d.prove(tb, input_claim=ob.Stage4Prover(steps, gamma, r_cycle).computeInputClaim())
    assert want.keys() == got.keys()
    for name in want:
        assert np.array_equal(want[name], got[name]), name
    assert ta.state == tb.state
    d.deinit()


def test_original_stage4_prover_rejects_nothing_it_should_not(api):
    """an inconsistent claim: the original prover's evaluations do not depend on it (p(1) comes from the tables)"""
    steps = seeded_steps(9, 40)
    r = ob.f_to_mont(ob.FR, np.random.default_rng(1).integers(0, 1 << 63, size=(8, 4), dtype=np.uint64))
    d = api.Stage4Prover(steps, r[0], r[1:7])
    a = d.computeRoundEvals(0, r[7])
    b = d.computeRoundEvals(0, api.fr_from_int(0))
    assert np.array_equal(a, b)
    d.deinit()
