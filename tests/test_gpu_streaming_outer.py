"""StreamingOuterProver's remaining rounds (src/zkvm/spartan/streaming_outer.zig) on the device: Az / Bz materialised by one affine-map
launch over the cycle witnesses (zg_fr_rows_affine), Gruen rounds and folds in a product session — against the restatement (oracle/)
and the reference's captured Stage-1 run."""
import json
import os

import numpy as np
import pytest

from oracle import binding as ob
from tests import util as U
from tests.test_transcript_host import check_stage1_outer_chain_of_the_captured_run, outer_true_claim, random_cycle_witnesses

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from zolt_amd import api, lib
    lib.init(0)
    return api


def test_captured_stage1_chain_with_the_mirrors_host_algebra(api, golden_dir):
    """the mirror's Lagrange weights / kernel, split-eq scalar (prefix tables from the device) and cubic on the reference's printed run"""
    fx = json.load(open(os.path.join(golden_dir, "stage1_outer_rounds.json")))
    check_stage1_outer_chain_of_the_captured_run(fx, api.GruenSplitEqPolynomial, api.lagrangeEvals, api.lagrangeKernel, api.fr_from_int, api.fr_to_int,
                                                 api.cubicAtPoint)


def test_constraint_table_matches_the_restatement(api):
    assert api.UNIFORM_CONSTRAINTS == ob.UNIFORM_CONSTRAINTS and list(api.FIRST_GROUP_INDICES) == ob.FIRST_GROUP_INDICES
    assert list(api.SECOND_GROUP_INDICES) == ob.SECOND_GROUP_INDICES and api.NUM_R1CS_INPUTS == ob.NUM_R1CS_INPUTS


@pytest.mark.parametrize("k,ntab,g,n_rows,n_pad", [(43, 2, 2, 100, 128), (1, 1, 1, 1, 1), (64, 4, 4, 65, 65), (7, 3, 1, 0, 4), (43, 2, 2, 5000, 8192)])
def test_rows_affine_against_integers(api, k, ntab, g, n_rows, n_pad):
    """zg_fr_rows_affine (host pointers) on random matrices with sparse coefficient rows, a zero row and a constant-only row"""
    from zolt_amd import lib
    rng = np.random.default_rng(k * 1000 + n_rows)
    rows = ob.f_to_mont(ob.FR, rng.integers(0, 1 << 63, size=(max(n_rows, 1) * k, 4), dtype=np.uint64)).reshape(-1, k, 4)[:n_rows]
    nout = ntab * g
    coeff_int = [[int(rng.integers(0, 1 << 62)) ** 4 % ob._R_P if rng.random() < 0.4 else 0 for _ in range(k + 1)] for _ in range(nout)]
    coeff_int[0] = [0] * (k + 1)
    if nout > 1:
        coeff_int[1] = [0] * k + [12345]
    coeffs = np.stack([np.stack([ob.fr_from_int(v) for v in row]) for row in coeff_int])
    outs = lib.fr_rows_affine(rows, coeffs, ntab, g, n_pad)
    assert len(outs) == ntab and all(o.shape == (n_pad * g, 4) for o in outs)
    check = sorted(set([0, n_rows // 2, max(n_rows - 1, 0)])) if n_rows else []
    for i in check:
        vals = [ob.fr_to_int(x) for x in rows[i]]
        for c in range(nout):
            want = (coeff_int[c][k] + sum(a * b for a, b in zip(coeff_int[c][:k], vals))) % ob._R_P
            assert ob.fr_to_int(outs[c // g][i * g + c % g]) == want, (i, c)
    for o in outs:
        assert not o[n_rows * g:].any()
    # the whole result against the vectorised oracle arithmetic
    if n_rows:
        for c in range(nout):
            acc = np.repeat(coeffs[c, k].reshape(1, 4), n_rows, axis=0)
            for col in range(k):
                if coeff_int[c][col]:
                    acc = ob._fadd(acc, ob._fmul(rows[:, col], coeffs[c, col]))
            assert np.array_equal(outs[c // g][c % g:n_rows * g:g], acc), c


@pytest.mark.parametrize("n_cycles", [1, 3, 256, 5000])
def test_rounds_against_the_restatement(api, n_cycles):
    """materialised Az / Bz, every round's (t'(0), t'(inf)) and four evaluations, the claim chain and the final point, bit for bit; trace
    lengths that are not powers of two, a single cycle"""
    w = random_cycle_witnesses(n_cycles + 7, n_cycles)
    T = 1
    while T < n_cycles:
        T *= 2
    nv = T.bit_length() - 1
    r = ob.f_to_mont(ob.FR, U.random_raw256(90 + n_cycles, 3 * nv + 8))
    tau, r0, scale, chals = r[:nv + 2], r[nv + 2], r[nv + 3], r[nv + 4:]
    o = ob.StreamingOuterProver(w, tau, scale)
    d = api.StreamingOuterProver(w, tau, scale)
    o.bindFirstRoundChallenge(r0, ob.fr_from_int(0))
    d.bindFirstRoundChallenge(r0, ob.fr_from_int(0))
    o.materializeLinearPhasePolynomials()
    d.materializeLinearPhasePolynomials()
    assert np.array_equal(d._s.read(0), o.az) and np.array_equal(d._s.read(1), o.bz)
    o.current_claim = outer_true_claim(o)
    d.current_claim = o.current_claim.copy()
    assert d.numRounds() == o.numRounds() == nv + 1
    for k in range(o.numRounds()):
        eo, ed = o.computeRemainingRoundPoly(), d.computeRemainingRoundPoly()
        assert np.array_equal(o.last_t[0], d.last_t[0]) and np.array_equal(o.last_t[1], d.last_t[1]), k
        assert np.array_equal(eo, ed), k
        assert np.array_equal(ob.f_add(ob.FR, eo[0:1], eo[1:2])[0], o.current_claim), k
        for p in (o, d):
            p.updateClaim(eo, chals[k])
            p.bindRemainingRoundChallenge(chals[k])
        assert np.array_equal(o.current_claim, d.current_claim)
    az, bz = d.finalAzBz()
    assert np.array_equal(az, o.az[0]) and np.array_equal(bz, o.bz[0])
    assert np.array_equal(ob._fmul(ob._fmul(az, bz), d.split_eq.current_scalar).reshape(4), d.getFinalEval())
    d.deinit()


def test_full_size_claim_chain(api):
    """2^20 cycles (1.4 GB of witnesses): the materialised tables against the restatement on a sample of cycles, then the size-independent
    property — from the true sum (a device dot product of eq, Az, Bz) every round's s(0) + s(1) is the claim and the last claim is
    scalar * Az * Bz at the bound point."""
    from zolt_amd import lib
    nv = 20
    n = (1 << nv) - 12345
    rng = np.random.default_rng(2)
    u = rng.integers(0, 1 << 63, size=(n, ob.NUM_R1CS_INPUTS), dtype=np.uint64)
    flags = [i for i, name in enumerate(ob.R1CS_INPUT_NAMES) if name.startswith(("Flag", "Should", "Write", "NextIs"))]
    u[:, flags] &= np.uint64(1)
    l = np.zeros((n * ob.NUM_R1CS_INPUTS, 4), dtype=np.uint64)
    l[:, 0] = u.reshape(-1)
    w = lib.field_op(lib.FR, lib.OP_TO_MONT, l).reshape(n, ob.NUM_R1CS_INPUTS, 4)
    del l, u
    r = ob.f_to_mont(ob.FR, U.random_raw256(77, 3 * nv + 8))
    tau, r0, scale, chals = r[:nv + 2], r[nv + 2], r[nv + 3], r[nv + 4:]
    d = api.StreamingOuterProver(w, tau, scale)
    d.bindFirstRoundChallenge(r0, api.fr_from_int(0))
    d.materializeLinearPhasePolynomials()
    az, bz = d._s.read(0), d._s.read(1)
    sample = np.concatenate([np.arange(0, 64), rng.integers(0, n, size=128), np.arange(n - 64, n)])
    o = ob.StreamingOuterProver(w[sample], tau[:10], scale)  # (only its per-cycle map is used: 256 cycles)
    o.bindFirstRoundChallenge(r0, ob.fr_from_int(0))
    o.materializeLinearPhasePolynomials()
    idx = np.stack([2 * sample, 2 * sample + 1], axis=1).reshape(-1)
    assert np.array_equal(az[idx], o.az) and np.array_equal(bz[idx], o.bz)
    assert not az[2 * n:].any() and not bz[2 * n:].any()
    eq = lib.fr_eq_table(tau[:-1], scale)
    d.current_claim = ob._fsum(ob._fmul(ob._fmul(az, bz), eq))
    del az, bz, eq
    for k in range(d.numRounds()):
        ev = d.computeRemainingRoundPoly()
        assert np.array_equal(ob.f_add(ob.FR, ev[0:1], ev[1:2])[0], d.current_claim), k
        d.updateClaim(ev, chals[k])
        d.bindRemainingRoundChallenge(chals[k])
    fa, fb = d.finalAzBz()
    assert np.array_equal(ob._fmul(ob._fmul(fa, fb), d.split_eq.current_scalar).reshape(4), d.getFinalEval())
    d.deinit()


@pytest.mark.parametrize("n_cycles", [1, 6, 256, 3000])
def test_uniskip_first_round_against_the_restatement(api, n_cycles):
    """computeFirstRoundPoly: t1 at the nine targets (one prodsum launch) and the 28 coefficients of s1 = L(tau_high, .) t1, bit for bit;
    s1 vanishes on the base window {-4..5} where t1 does (constraint products of a SATISFYING assignment would — random inputs only give
    t1's zeros by construction of t1_vals), so s1(0) = 0: coefficient 0 is zero, as the captured run prints (logs/zolt.log:1661)"""
    w = random_cycle_witnesses(n_cycles + 11, n_cycles)
    nv = max(n_cycles - 1, 0).bit_length()
    r = ob.f_to_mont(ob.FR, U.random_raw256(190 + n_cycles, nv + 3))
    tau, scale = r[:nv + 2], r[nv + 2]
    o = ob.StreamingOuterProver(w, tau, scale)
    d = api.StreamingOuterProver(w, tau, scale)
    want, got = o.computeFirstRoundPoly(), d.computeFirstRoundPoly()
    assert o.last_extended_evals == d.last_extended_evals
    assert np.array_equal(want, got) and got.shape == (28, 4)
    assert not got[0].any()
    # s1 at a target equals L(tau_high, target) * t1(target), t1(target) the device sum
    P = ob._R_P
    co = [ob.fr_to_int(x) for x in got]
    for z, t1z in zip(api.UNISKIP_TARGETS, d.last_extended_evals):
        lag = [ob.fr_to_int(x) for x in ob.lagrange_evals_symmetric(tau[-1], 10)]
        basis_at_z = [ob.fr_to_int(x) for x in ob.lagrange_evals_symmetric(ob.fr_from_int(z % P), 10)]
        kernel = sum(a * b for a, b in zip(lag, basis_at_z)) % P
        assert sum(c * pow(z, k, P) for k, c in enumerate(co)) % P == kernel * t1z % P
    d.deinit()


def test_uniskip_constants(api):
    assert api.UNISKIP_TARGETS == ob.UNISKIP_TARGETS == [-5, 6, -6, 7, -7, 8, -8, 9, -9] and api.COEFFS_PER_J == ob.COEFFS_PER_J
    for j, t in enumerate(api.UNISKIP_TARGETS):  # the shift coefficients ARE the Lagrange basis of {-4..5} at the target
        assert [c % ob._R_P for c in api.COEFFS_PER_J[j]] == [ob.fr_to_int(x) for x in api.lagrangeEvals(api.fr_from_int(t % ob._R_P), 10)]


def test_stage1_of_the_captured_run_on_the_device(api, golden_dir):
    """Stage 1 of the reference's captured run through the DEVICE path, from the witnesses regenerated out of the ELF: the 36 R1CS input
    claims (zg_fr_rows_mle over the cycle-major matrix), the UniSkip first-round polynomial (zg_fr_rows_affine_prodsum_dev) and the nine
    rounds (zg_fr_rows_affine_dev + the product session) — every printed value, full width."""
    from zolt_amd import lib
    from tests.test_transcript_host import (check_r1cs_claims_of_the_captured_run, check_stage1_outer_against_the_captured_run,
                                            stage1_witness_of_the_captured_run)
    w = stage1_witness_of_the_captured_run(golden_dir)
    cl = json.load(open(os.path.join(golden_dir, "stage1_r1cs_claims.json")))
    check_r1cs_claims_of_the_captured_run(lambda r: lib.fr_rows_mle(w, r), cl)
    fx = json.load(open(os.path.join(golden_dir, "stage1_outer_rounds.json")))
    provers = []

    def make(tau, scale):
        provers.append(api.StreamingOuterProver(w, tau, scale))
        return provers[-1]
    check_stage1_outer_against_the_captured_run(make, fx, api.fr_from_int, api.fr_to_int, api.lagrangeKernel)
    for p in provers:
        p.deinit()


@pytest.mark.parametrize("small_path,stage", [("1", "1"), ("1", "0"), ("0", "1")])
def test_affine_product_sums_with_small_and_general_coefficients(api, small_path, stage, monkeypatch):
    """zg_fr_rows_affine_prodsum_dev term by term against Python integers: coefficients that are small integers (+-m, m < 2^24: an
    8 x 1-limb product summed as a 288-bit integer, one reduction per map), the values on either side of that bound, full-width ones
    (the general product), all in one map; 64 terms of the largest small magnitude on rows of r - 1 (the bound of the integer sum); a
    zero constant and a full-width constant. ZG_ROWS_SMALL_COEFF=0 sends every term through the general product, ZG_ROWS_STAGE=0 reads
    the rows from memory per term instead of from the workgroup's tile in LDS: same bytes."""
    from zolt_amd import lib
    monkeypatch.setenv("ZG_ROWS_SMALL_COEFF", small_path)
    monkeypatch.setenv("ZG_ROWS_STAGE", stage)
    P = ob._R_P
    rng = np.random.default_rng(5)
    k, n_rows, npairs, g = 70, 300, 7, 2  # seven pairs: a full group of four and a short one
    rows_int = [[int(x) for x in rng.integers(0, 1 << 62, size=k)] for _ in range(n_rows)]
    for i in range(0, n_rows, 7):
        rows_int[i] = [P - 1] * k  # the largest stored element in every column
    rows_int[3] = [0] * k
    big = [int.from_bytes(rng.bytes(32), "little") % P for _ in range(40)]
    kinds = [1, P - 1, 5, P - 7, (1 << 24) - 1, P - ((1 << 24) - 1), 1 << 24, P - (1 << 24), (1 << 24) + 1, 12345, P - 99999] + big
    coeff_int = []
    for c in range(2 * npairs):
        row = [0] * (k + 1)
        if c == 0:  # 64 terms, every one the largest small magnitude, alternating sign
            for col in range(64):
                row[col] = (1 << 24) - 1 if col % 2 == 0 else P - ((1 << 24) - 1)
        elif c == 1:  # 64 small positive terms of the largest magnitude: the bound of the positive sum
            for col in range(64):
                row[col] = (1 << 24) - 1
        else:
            cols = rng.choice(k, size=int(rng.integers(1, 50)), replace=False)
            for col in cols:
                row[int(col)] = kinds[int(rng.integers(0, len(kinds)))]
        row[k] = 0 if c % 3 == 0 else big[c]
        coeff_int.append(row)
    w_int = [int.from_bytes(rng.bytes(32), "little") % P for _ in range(n_rows * g)]

    def mont(vals):
        raw = np.array([[(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)] for v in vals], dtype=np.uint64)
        return ob.f_to_mont(ob.FR, raw)
    d_rows = lib.DeviceBuffer.from_host(mont([x for r in rows_int for x in r]))
    d_w = lib.DeviceBuffer.from_host(mont(w_int))
    got = lib.fr_rows_affine_prodsum_dev(d_rows.ptr, n_rows, k, mont([x for r in coeff_int for x in r]), npairs, d_w.ptr, g)
    for p in range(npairs):
        a, b = coeff_int[2 * p], coeff_int[2 * p + 1]
        tot = 0
        for i in range(n_rows):
            av = (a[k] + sum(a[c] * rows_int[i][c] for c in range(k))) % P
            bv = (b[k] + sum(b[c] * rows_int[i][c] for c in range(k))) % P
            tot += w_int[i * g + p % g] * av * bv
        assert ob.fr_to_int(got[p]) == tot % P, (p, small_path, stage)
