#!/usr/bin/env python3
"""Randomised differential test of the poly / sumcheck entry points against the CPU oracle.  usage: fuzz_poly.py [seconds=60] [seed=1]"""
import ctypes as C
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import binding as ob  # checker
from zolt_amd import api, lib

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
lib.init(0)
lib.init_devices(1)


def rand_fr(n, sparse=False):
    raw = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    a = ob.f_to_mont(ob.FR, raw)
    if sparse and n:
        a[rng.random(n) < 0.5] = 0
    return a


def rand_ch(k=None):
    """challenges: half of them in the reference's stored 128-bit form [0, 0, lo, hi] (the fold kernels' narrow product)"""
    a = rand_fr(1 if k is None else k)
    for row in a:
        if rng.random() < 0.5:
            row[:2] = 0
            row[3] &= np.uint64((1 << 61) - 1)
    return a[0] if k is None else a


def dev(a):
    p = C.c_void_p()
    assert lib._lib.zg_dev_alloc(C.c_size_t(max(a.size, 1) * 8), C.byref(p)) == 0
    if a.size:
        assert lib._lib.zg_memcpy_h2d(p, np.ascontiguousarray(a).ctypes.data_as(C.c_void_p), C.c_size_t(a.size * 8)) == 0
    return p


t0 = time.time()
cases = 0
while time.time() - t0 < budget:
    v = int(rng.integers(0, 15))
    n = 1 << v
    sparse = rng.random() < 0.3
    tab = rand_fr(n, sparse)
    # device-resident runSumcheck vs oracle
    res = lib.run_sumcheck(tab)
    wc, wr, wch, wfin, wok = ob.run_sumcheck(tab)
    assert np.array_equal(res["claim"], wc) and np.array_equal(res["final_eval"], wfin) and res["result"] == bool(wok)
    assert np.array_equal(res["rounds"].reshape(-1, 2, 4), np.asarray(wr).reshape(-1, 2, 4))
    # sessions in both layouts with arbitrary challenges
    for layout in (0, 1):
        s = lib.SumcheckSession.open(tab, layout)
        cur = tab
        for k in range(v):
            g0, g1 = s.round_sums()
            w0, w1 = ob.fr_sum_halves(cur) if layout == 0 else ob.fr_sum_even_odd(cur)
            assert np.array_equal(g0, w0) and np.array_equal(g1, w1)
            ch = rand_ch()
            s.bind(ch)
            cur = ob.fr_bind_high(cur, ch) if layout == 0 else ob.fr_bind_low(cur, ch)
        assert np.array_equal(s.final(), cur[0])
        s.close()
    # eq table (scaled or not), Spartan combine, fused opening
    r = rand_fr(v)
    scale = rand_fr(1)[0] if rng.random() < 0.5 else None
    eq = lib.fr_eq_table(r, scale)
    assert np.array_equal(eq, ob.fr_eq_table(r, scale))
    if v <= 12:  # GruenSplitEqPolynomial's prefix-table set of the same point
        for a, b in zip(lib.fr_eq_prefix_tables(r), ob.fr_eq_prefix_tables(r)):
            assert np.array_equal(a, b), ("prefix tables", v)
    az, bz, cz = rand_fr(n, sparse), rand_fr(n), rand_fr(n, sparse)
    f = lib.fr_spartan_combine(eq, az, bz, cz)
    assert np.array_equal(f, ob.fr_spartan_combine(eq, az, bz, cz))
    d = [dev(x) for x in (az, bz, cz)]
    layout = int(rng.integers(0, 2))
    s = lib.SumcheckSession.open_spartan_dev(r, d[0].value, d[1].value, d[2].value, layout=layout, scale=scale)
    assert np.array_equal(s.read(), f)
    if v:
        g0, g1 = s.round_sums()
        w0, w1 = ob.fr_sum_halves(f) if layout == 0 else ob.fr_sum_even_odd(f)
        assert np.array_equal(g0, w0) and np.array_equal(g1, w1)
    s.close()
    for p in d:
        lib._lib.zg_dev_free(p)
    if v <= 12:
        pt = rand_fr(v)
        assert np.array_equal(lib.fr_dense_evaluate(tab, pt), ob.fr_dense_evaluate(tab, pt))
    # round-2 prover sites: RAF cubic round sums on a LOW_PAIR session, Lasso bit-split sums, a transcript-driven Stage-1 loop,
    # DensePolynomial.scale, and the sharded session with a random number of logical shards
    if v >= 1:
        s = lib.SumcheckSession.open(tab, lib.SC_LOW_PAIR)
        start = int(rng.integers(0, 1 << 40))
        claim = rand_fr(1)[0]
        bound = np.zeros((0, 4), dtype=np.uint64)
        cur = tab
        for k in range(min(v, 4)):
            from zolt_amd import api
            base = start % api.R_MOD
            power = 8
            for bv in bound:
                base = (base + api.fr_to_int(bv) * power) % api.R_MOD
                power *= 2
            s0, s2 = s.raf_round(api.fr_from_int(base), power)
            want = ob.raf_round_cubic(cur, start, bound, v, claim)
            assert np.array_equal(s0, want[0]) and np.array_equal(s2, want[2]), ("raf", v, k)
            ch = rand_ch()
            s.bind(ch)
            cur = ob.fr_bind_low(cur, ch)
            bound = np.concatenate([bound, ch[None, :]])
        s.close()
    idx = rng.integers(0, 1 << 63, size=(n, 2), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 2), dtype=np.uint64)
    bit = int(rng.integers(0, 128))
    h0, h1 = lib.fr_bit_split_sums(tab, idx, bit)
    w0, w1 = ob.lasso_address_sums(tab, idx, bit)
    assert np.array_equal(h0, w0) and np.array_equal(h1, w1), ("bit split", v, bit)
    from zolt_amd import api
    rounds = v + int(rng.integers(0, 2))
    got = api.proveStage1(tab, rounds, api.Transcript(b"Jolt"))
    wrp, wch, wfin = ob.stage1_prove(tab, rounds, ob.Transcript(b"Jolt"))
    assert np.array_equal(got["round_polys"], wrp) and np.array_equal(got["challenges"], wch) and np.array_equal(got["final_eval"], wfin), ("stage1", v)
    sc1 = rand_fr(1)[0]
    assert np.array_equal(lib.fr_scale(tab, sc1), ob.fr_poly_scale(tab, sc1))
    os.environ["ZG_SHARDS"] = str(int(rng.integers(1, 9)))
    layout = int(rng.integers(0, 2))
    ss = lib.ShardedSumcheckSession.open(tab, layout)
    cur = tab
    for k in range(v):
        g0, g1 = ss.round_sums()
        w0, w1 = ob.fr_sum_halves(cur) if layout == 0 else ob.fr_sum_even_odd(cur)
        assert np.array_equal(g0, w0) and np.array_equal(g1, w1), ("sharded session", v, k)
        ch = rand_ch()
        ss.bind(ch)
        cur = ob.fr_bind_high(cur, ch) if layout == 0 else ob.fr_bind_low(cur, ch)
    assert np.array_equal(ss.final(), cur[0])
    ss.close()
    # product-form provers on one k-table session against the oracle's restatement of each loop
    if v >= 1:
        kind = int(rng.integers(0, 4))
        chs = rand_ch(v)
        claim = rand_fr(1)[0]
        if kind <= 1:
            tabs = [rand_fr(n, sparse), rand_fr(n), rand_fr(n) if kind == 0 else None]
            g = api.ValEvaluationProver(*tabs, claim) if kind == 0 else api.ValFinalProver(tabs[0], tabs[1], claim)
            o = ob.ValEvaluationProver(*tabs, claim)
            for k in range(v):
                rp, wrp = g.computeRoundPolynomial(), o.computeRoundPolynomial()
                assert np.array_equal(rp, wrp), ("val", kind, v, k)
                g.bindChallengeWithPoly(chs[k], rp)
                o.bindChallengeWithPoly(chs[k], wrp)
                assert np.array_equal(g.current_claim, o.current_claim)
            assert all(np.array_equal(a, b) for a, b in zip(g.getFinalClaims(), o.getFinalClaims()))
        elif kind == 2:
            tabs = [rand_fr(n, sparse and j == 1) for j in range(5)]
            g, o = api.OutputSumcheckProver(*tabs, claim), ob.OutputSumcheckProver(*tabs, claim)
            for k in range(v):
                ev, wev = g.roundEvals(), o.roundEvals()
                assert np.array_equal(ev, wev), ("output", v, k)
                g.bindChallenge(chs[k]); o.bindChallenge(chs[k])
                g.updateClaim(ev, chs[k]); o.updateClaim(wev, chs[k])
                assert np.array_equal(g.current_claim, o.current_claim)
            fg, fo = g.getFinalClaims(), o.getFinalClaims()
            assert all(np.array_equal(fg[key], fo[key]) for key in fo)
        else:
            left, right, tau, kern = rand_fr(n, sparse), rand_fr(n), rand_fr(v), rand_fr(1)[0]
            g, o = api.ProductVirtualRemainderProver(left, right, tau, kern, claim), ob.ProductRemainderProver(left, right, tau, kern, claim)
            for k in range(v):
                ev, wev = g.roundEvals(), o.roundEvals()
                assert np.array_equal(ev, wev), ("product", v, k)
                g.bindChallenge(chs[k]); o.bindChallenge(chs[k])
                g.updateClaim(ev, chs[k]); o.updateClaim(wev, chs[k])
            assert np.array_equal(g.getFinalClaim(), o.getFinalClaim())
        g.deinit()
    # Stage-3 shapes: InstructionInput (four terms over ten tables) and Shift phase 2 (three terms over seven)
    if 1 <= v <= 11:
        gamma = rand_fr(1)[0]
        tabs = [rand_fr(n, sparse and j % 2 == 0) for j in range(10)]
        g = api.InstructionInputProver(tabs, gamma)
        cur, claim = [t.copy() for t in tabs], rand_fr(1)[0]
        for k in range(min(v, 5)):
            want = ob.instruction_input_round(cur, gamma, claim)
            assert np.array_equal(g.computeRoundEvals(claim), want), ("instruction input", v, k)
            ch = rand_ch()
            g.bind(ch)
            cur = [ob.fr_bind_low(t, ch) for t in cur]
            claim = ob.raf_update_claim(want, ch)
        g.deinit()
        gp = rand_fr(5)
        tabs = [rand_fr(n, sparse and j == 6) for j in range(7)]
        g = api.ShiftSumcheckRounds(tabs, phase2=True, gamma_powers=gp)
        cur = [t.copy() for t in tabs]
        for k in range(min(v, 5)):
            assert np.array_equal(g.computeRoundEvals(claim), ob.shift_phase2_round(cur, gp, claim)), ("shift phase 2", v, k)
            ch = rand_ch()
            g.bind(ch)
            cur = [ob.fr_bind_low(t, ch) for t in cur]
        g.deinit()
    # LassoProver with a ragged cycle count
    if v <= 10:
        log_K = int(rng.integers(1, 129))
        ncyc = int(rng.integers(1, n + 1))
        lidx = rng.integers(0, 1 << 63, size=(ncyc, 2), dtype=np.uint64)
        if log_K < 128:
            mask = (1 << log_K) - 1
            lidx[:, 0] &= np.uint64(mask & (2**64 - 1))
            lidx[:, 1] &= np.uint64(mask >> 64)
        w = rand_fr(v)
        got, want = api.runLassoProver(lidx, v, log_K, w), ob.run_lasso_prover(lidx, v, log_K, w)
        assert all(np.array_equal(got[key], want[key]) for key in want), ("lasso", v, log_K, ncyc)
    # weighted column sums and Dory's vector-matrix product / evaluation vectors
    if v >= 1:
        rows, m = 1 << int(rng.integers(0, v + 1)), int(rng.integers(1, 5))
        cols = n // rows
        tab, wts = rand_fr(n, sparse), rand_fr(m * rows).reshape(m, rows, 4)
        got = lib.fr_weighted_colsum(tab, rows, cols, wts)
        t3 = tab.reshape(rows, cols, 4)
        for c in {0, cols - 1, int(rng.integers(0, cols))}:
            for k in range(m):
                assert np.array_equal(got[k, c], ob._fsum(ob._fmul(np.ascontiguousarray(t3[:, c]), np.ascontiguousarray(wts[k])))), ("colsum", rows, cols, k, c)
    if v <= 8:
        nu, sigma = int(rng.integers(0, 4)), int(rng.integers(0, 4))
        ne, nl = int(rng.integers(1, (1 << (nu + sigma)) + 3)), int(rng.integers(1, (1 << nu) + 2))
        ev, lv = rand_fr(ne, sparse), rand_fr(nl)
        assert np.array_equal(api.Dory.computeVectorMatrixProduct(ev, lv, nu, sigma), ob.dory_vector_matrix_product(ev, lv, nu, sigma)), ("dory vmp", nu, sigma, ne, nl)
        pt = rand_fr(int(rng.integers(0, nu + sigma + 3)))
        gl, gr = api.Dory.computeEvaluationVectors(pt, nu, sigma)
        wl, wr = ob.dory_evaluation_vectors(pt, nu, sigma)
        assert np.array_equal(gl, wl) and np.array_equal(gr, wr), ("dory vectors", nu, sigma, pt.shape[0])
    # Stage 3's prefix / suffix provers as a whole, over random padded witnesses
    if 2 <= v <= 7:
        wm = rand_fr(n * 43, sparse).reshape(n, 43, 4)
        wi = [[ob.fr_to_int(x) for x in row] for row in wm]
        ro, rp, g3 = rand_fr(v), rand_fr(v), rand_fr(2)
        gi = ob.fr_to_int(g3[0])
        sgi = [pow(gi, i, ob._R_P) for i in range(5)]
        want = (ob.Stage3ShiftProver(wi, [ob.fr_to_int(x) for x in ro], [ob.fr_to_int(x) for x in rp], sgi),
                ob.Stage3RegistersProver(wi, [ob.fr_to_int(x) for x in ro], ob.fr_to_int(g3[1])))
        gotp = (api.ShiftPrefixSuffixProver(wm, ro, rp, np.stack([ob.fr_from_int(x) for x in sgi])), api.RegistersPrefixSuffixProver(wm, ro, g3[1]))
        claim = rand_fr(1)[0]
        for k in range(v):
            ch = rand_ch()
            for a, b in zip(gotp, want):
                assert np.array_equal(a.computeRoundEvals(claim), b.computeRoundEvals(claim)), ("stage3", v, k, type(a).__name__)
                a.bind(ch)
                b.bind(ob.fr_to_int(ch))
        for a in gotp:
            a.deinit()
    cases += 1
print(f"fuzz ok: {cases} random instances (all entry points) in {time.time() - t0:.1f} s")
