"""The Python host mirror (zolt_amd/api/) against the oracle: HyperKZG setup/commit/batchCommit/open, the
polynomial classes, runSumcheck; plus re-entrancy of the C ABI from several host threads."""
import threading

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


def test_hyperkzg_setup_commit_open(env):
    """src/poly/commitment/mod.zig:174-324: mock SRS, commit, batchCommit, open (quotient commitments + final eval)."""
    api, lib, ob = env
    params = api.HyperKZG.setup(64)
    wsrs, winf = ob.hyperkzg_setup(64)
    assert np.array_equal(params.powers_of_tau_g1, wsrs) and np.array_equal(params.infinity, winf)
    evals = _rand(ob, 1, 64)
    c, ci = api.HyperKZG.commit(params, evals)
    wc, wci = ob.hyperkzg_commit(wsrs, winf, evals)
    assert ci == wci and np.array_equal(c, wc)
    # evals longer than the SRS are truncated to n = min(len, srs) (:246); empty -> identity (:240-242)
    c2, _ = api.HyperKZG.commit(params, _rand(ob, 2, 100))
    w2, _ = ob.hyperkzg_commit(wsrs, winf, _rand(ob, 2, 100))
    assert np.array_equal(c2, w2)
    z, zi = api.HyperKZG.commit(params, np.zeros((0, 4), dtype=np.uint64))
    assert zi == 1 and not z.any() and api.commitment_to_bytes(z, zi) == bytes(64)
    # equal lengths are fused into one launch set; mixed lengths (short, empty, longer than the SRS) keep their order
    polys = [_rand(ob, 10 + k, 64) for k in range(3)] + [_rand(ob, 20, 10), np.zeros((0, 4), dtype=np.uint64), _rand(ob, 21, 100),
                                                         _rand(ob, 22, 10)]
    for (bc, bi), p in zip(api.HyperKZG.batchCommit(params, polys), polys):
        w, wi = ob.hyperkzg_commit(wsrs, winf, p)
        assert bi == wi and np.array_equal(bc, w)
    point = _rand(ob, 3, 6)
    quotients, final = api.HyperKZG.open(params, evals, point, np.zeros(4, dtype=np.uint64))
    wq, wqi, wfin = ob.hyperkzg_open(wsrs, winf, evals, point, np.zeros(4, dtype=np.uint64))
    assert np.array_equal(final, wfin) and len(quotients) == 6
    for i, (q, qi) in enumerate(quotients):
        assert qi == wqi[i] and np.array_equal(q, wq[i])
    # table shorter than 2^num_vars (folding stops, :289), odd length, SRS shorter than the quotient
    for n_ev, v, n_srs in ((5, 4, 64), (37, 5, 64), (64, 6, 20), (1, 3, 64), (128, 7, 64)):
        ev = _rand(ob, 40 + n_ev, n_ev)
        pt = _rand(ob, 50 + v, v)
        pr = api.HyperKZG.setup(n_srs)
        ws, wi = ob.hyperkzg_setup(n_srs)
        qs, fin = api.HyperKZG.open(pr, ev, pt, np.zeros(4, dtype=np.uint64))
        wq, wqi, wfin = ob.hyperkzg_open(ws, wi, ev, pt, np.zeros(4, dtype=np.uint64))
        assert np.array_equal(fin, wfin), (n_ev, v)
        done = 0
        ln = n_ev
        while done < v and ln // 2 > 0:   # rounds the reference actually executes
            assert qs[done][1] == wqi[done] and np.array_equal(qs[done][0], wq[done]), (n_ev, v, done)
            ln //= 2
            done += 1
        pr.deinit()
    params.deinit()


def test_hyperkzg_key_planned_for_one_proof(env):
    """HyperKZG.setup(n, expected_uses = 1): the key of a prover that proves once (no table of multiples) — the same powers of tau, and
    the same commitments (field elements and machine words), batch commitments and opening as the key that lives on, at a size where
    the two handles take different plans (20 000 powers: 17-bit windows with a table against table-less windows + the side table)."""
    api, lib, ob = env
    n = 20000
    once, keeps = api.HyperKZG.setup(n, expected_uses=1), api.HyperKZG.setup(n)
    assert np.array_equal(once.powers_of_tau_g1, keeps.powers_of_tau_g1) and np.array_equal(once.infinity, keeps.infinity)
    assert once._dev.plan()[2] == 1 and keeps._dev.plan()[2] > 1
    ev = _rand(ob, 1201, n)
    want = ob.hyperkzg_commit(keeps.powers_of_tau_g1, keeps.infinity, ev)
    for params in (once, keeps):
        c, ci = api.HyperKZG.commit(params, ev)
        assert ci == want[1] and np.array_equal(c, want[0])
    words = U.splitmix64(1202, n)
    a, b = api.HyperKZG.commitU64(once, words), api.HyperKZG.commitU64(keeps, words)
    assert a[1] == b[1] and np.array_equal(a[0], b[0])
    polys = [_rand(ob, 1210 + k, n) for k in range(3)]
    for (x, xi), (y, yi) in zip(api.HyperKZG.batchCommit(once, polys), api.HyperKZG.batchCommit(keeps, polys)):
        assert xi == yi and np.array_equal(x, y)
    v = 14
    ev = _rand(ob, 1203, 1 << v)
    pt = _rand(ob, 1204, v)
    q1, f1 = api.HyperKZG.open(once, ev, pt, np.zeros(4, dtype=np.uint64))
    q2, f2 = api.HyperKZG.open(keeps, ev, pt, np.zeros(4, dtype=np.uint64))
    assert np.array_equal(f1, f2) and all(i1 == i2 and np.array_equal(x1, x2) for (x1, i1), (x2, i2) in zip(q1, q2))
    once.deinit()
    keeps.deinit()


@pytest.mark.parametrize("srs_n,lens,v", [(64, [64, 64, 64], 6), (64, [32, 16, 40, 0], 5), (16, [8], 3), (64, [64, 64], 0),
                                          (2048, [2048, 2048, 1000], 11), (64, [4, 4], 5), (64, [], 3), (2048, [1024, 1024], 10)])
def test_hyperkzg_batch_open(env, srs_n, lens, v):
    """HyperKZG.batchOpen (src/poly/commitment/mod.zig:607-732) incl. the reference's evaluateMultilinear (:788-817: direct sum
    only for point.len <= 10 and len <= 1024, evals[0] otherwise), shorter / longer / empty polynomials, zero variables,
    a fold that runs out of elements (fewer quotients than variables) and the empty batch."""
    api, lib, ob = env
    params = api.HyperKZG.setup(srs_n)
    polys = [_rand(ob, 300 + i, n) if n else np.zeros((0, 4), dtype=np.uint64) for i, n in enumerate(lens)]
    point = _rand(ob, 350, v) if v else np.zeros((0, 4), dtype=np.uint64)
    got = api.HyperKZG.batchOpen(params, polys, point)
    wq, wqi, wev, wfin, wgam = ob.hyperkzg_batch_open(params.powers_of_tau_g1, params.infinity, polys, point)
    assert len(got["quotient_commitments"]) == wq.shape[0]
    for i, (q, qi) in enumerate(got["quotient_commitments"]):
        assert qi == wqi[i] and np.array_equal(q, wq[i])
    assert np.array_equal(got["evaluations"], wev) and np.array_equal(got["final_eval"], wfin)
    assert np.array_equal(got["batching_challenge"], wgam)
    params.deinit()


@pytest.mark.parametrize("v,srs_n", [(13, 8192), (15, 32768), (12, 1000), (14, 16384)])
def test_hyperkzg_open_mid_sizes(env, v, srs_n):
    """open() at sizes where the short quotient commits are batched: a small-window SRS handle fuses them into one launch set
    (v = 13, 14), a 2^15-point handle (16-bit windows) takes the unfused rotation, and an SRS shorter than the table clamps
    every commit to its length (src/poly/commitment/mod.zig:246). Quotients and final evaluation equal the oracle's."""
    api, lib, ob = env
    gm = ob.g1_gen_multiples(srs_n)
    inf = np.zeros(srs_n, dtype=np.uint8)
    params = api.HyperKZG.SetupParams(gm, inf)
    ev = _rand(ob, 600 + v, 1 << v)
    pt = _rand(ob, 650 + v, v)
    quotients, final = api.HyperKZG.open(params, ev, pt, np.zeros(4, dtype=np.uint64))
    wq, wqi, wfin = ob.hyperkzg_open(gm, inf, ev, pt, np.zeros(4, dtype=np.uint64))
    assert np.array_equal(final, wfin) and len(quotients) == v
    for i, (q, qi) in enumerate(quotients):
        assert qi == wqi[i] and np.array_equal(q, wq[i]), i
    params.deinit()


@pytest.mark.parametrize("v,srs_n,fuse", [(17, 1 << 17, "2"), (18, 1 << 17, "2"), (18, 1 << 18, "2"), (18, 1 << 17, "1"), (17, 1 << 17, "1"), (18, 1 << 18, "1"), (17, 1 << 17, "0"),
                                          (18, 1 << 17, "1:single_pass_sort"), (17, 1 << 17, "2:single_pass_sort"), (18, 1 << 18, "1:fine_bits_min_8")])
def test_hyperkzg_open_long_levels(env, v, srs_n, fuse, monkeypatch):
    """open() with several long levels (quotients of more than 16384 entries) on a wide-window SRS handle: they are committed as
    the rows of one zero-padded matrix by ONE fused launch set (two-pass sort over several vectors' bucket sets), the short
    levels by another; v = 18 on a 2^17-point SRS clamps the first commit to the SRS length. ZG_HK_FUSE_LONG=1 (default): all long
    levels in the matrix; 2: the first long level keeps its own launch set and the matrix holds the rest at the second level's length
    (needs three long levels); 0: one launch set per long level on the helper streams. Same quotient commitments and final evaluation as the oracle either way."""
    api, lib, ob = env
    # (round 6) "...:single_pass_sort" / "...:fine_bits_min_8": the fused set of long levels is priced by the two-pass sort's coarse bins; with
    # that sort switched off (or refused for lack of fine key bits) the set must NOT be fused — it ran the single-pass LDS scatter over 160 k
    # counters and faulted. Found by running the whole suite under the alternate switches.
    fuse, _, alt = fuse.partition(":")
    if alt == "single_pass_sort":
        monkeypatch.setenv("ZG_MSM_TWO_PASS_SORT", "0")
    elif alt == "fine_bits_min_8":
        monkeypatch.setenv("ZG_MSM_FINE_BITS_MIN", "8")
    monkeypatch.setenv("ZG_HK_FUSE_LONG", fuse)
    gm = ob.g1_gen_multiples(srs_n)
    inf = np.zeros(srs_n, dtype=np.uint8)
    inf[7::1000] = 1
    params = api.HyperKZG.SetupParams(gm, inf)
    ev = _rand(ob, 700 + v, 1 << v)
    pt = _rand(ob, 750 + v, v)
    quotients, final = api.HyperKZG.open(params, ev, pt, np.zeros(4, dtype=np.uint64))
    wq, wqi, wfin = ob.hyperkzg_open(gm, inf, ev, pt, np.zeros(4, dtype=np.uint64))
    assert np.array_equal(final, wfin) and len(quotients) == v
    for i, (q, qi) in enumerate(quotients):
        assert qi == wqi[i] and np.array_equal(q, wq[i]), i
    # a point with fewer variables than the table: the fold stops after the long levels and ONE level of at most 16384
    # entries, which then has nothing to fuse with
    short = v - 14
    quotients, final = api.HyperKZG.open(params, ev, pt[:short], np.zeros(4, dtype=np.uint64))
    wq, wqi, wfin = ob.hyperkzg_open(gm, inf, ev, pt[:short], np.zeros(4, dtype=np.uint64))
    assert np.array_equal(final, wfin) and len(quotients) == short
    for i, (q, qi) in enumerate(quotients):
        assert qi == wqi[i] and np.array_equal(q, wq[i]), i
    params.deinit()


@pytest.mark.parametrize("v,srs_n,uses,shared", [(18, 1 << 18, 1, "1"), (18, 1 << 18, 3, "1"), (18, 1 << 17, 1, "1"), (17, 1 << 17, 1, "0"), (19, 1 << 19, 2, "1")])
def test_hyperkzg_open_on_a_key_without_a_table(env, v, srs_n, uses, shared, monkeypatch):
    """open() on a handle planned for a few MSMs (zg_msm_config.expected_uses = 1..15: no table of multiples, one bucket set per window):
    the long levels are sorted and accumulated one by one and reduced TOGETHER (msm_rows_shared_tail: one row / column pass, one msm_final,
    one window-combine launch for all of them; ZG_MSM_ROWS_SHARED_TAIL=0: a reduction per level) — same quotient commitments and final
    evaluation as the oracle, and as the batch entry point gives for the same rows."""
    api, lib, ob = env
    monkeypatch.setenv("ZG_MSM_ROWS_SHARED_TAIL", shared)
    gm = ob.g1_gen_multiples(srs_n)
    inf = np.zeros(srs_n, dtype=np.uint8)
    inf[5::777] = 1
    dev = lib.Bases.upload(gm, inf, expected_uses=uses)
    c, w, levels = dev.plan()
    assert levels == 1 and w > 1
    params = api.HyperKZG.SetupParams(gm, inf, dev=dev)
    ev = _rand(ob, 900 + v, 1 << v)
    pt = _rand(ob, 950 + v, v)
    for _ in range(2):  # the second call reuses the rows' buffers
        quotients, final = api.HyperKZG.open(params, ev, pt, np.zeros(4, dtype=np.uint64))
        wq, wqi, wfin = ob.hyperkzg_open(gm, inf, ev, pt, np.zeros(4, dtype=np.uint64))
        assert np.array_equal(final, wfin) and len(quotients) == v
        for i, (q, qi) in enumerate(quotients):
            assert qi == wqi[i] and np.array_equal(q, wq[i]), i
    params.deinit()


def test_hyperkzg_open_at_the_bench_size(env):
    """open() of 2^20 evaluations on a 2^20-point SRS — the size bench.py times — quotient by quotient against the oracle
    (the CPU side is ~2^20 points of Pippenger: tens of seconds)."""
    api, lib, ob = env
    v = 20
    gm = ob.g1_gen_multiples(1 << v)
    inf = np.zeros(1 << v, dtype=np.uint8)
    params = api.HyperKZG.SetupParams(gm, inf)
    ev = _rand(ob, 820, 1 << v)
    pt = _rand(ob, 821, v)
    quotients, final = api.HyperKZG.open(params, ev, pt, np.zeros(4, dtype=np.uint64))
    wq, wqi, wfin = ob.hyperkzg_open(gm, inf, ev, pt, np.zeros(4, dtype=np.uint64))
    assert np.array_equal(final, wfin) and len(quotients) == v
    for i, (q, qi) in enumerate(quotients):
        assert qi == wqi[i] and np.array_equal(q, wq[i]), i
    params.deinit()


def test_hyperkzg_open_resident_table(env):
    """zg_hyperkzg_open_dev: same proof as the host-table entry point, and the caller's device table is left intact."""
    api, lib, ob = env
    v = 15
    gm = ob.g1_gen_multiples(1 << v)
    params = api.HyperKZG.SetupParams(gm, np.zeros(1 << v, dtype=np.uint8))
    ev = _rand(ob, 840, 1 << v)
    pt = _rand(ob, 841, v)
    d_ev = lib.DeviceBuffer.from_host(ev)
    q, qi, fin = lib.hyperkzg_open_dev(params._dev, d_ev.ptr, 1 << v, pt, np.zeros(4, dtype=np.uint64))
    wq, wqi, wfin = ob.hyperkzg_open(gm, np.zeros(1 << v, dtype=np.uint8), ev, pt, np.zeros(4, dtype=np.uint64))
    assert np.array_equal(fin, wfin) and np.array_equal(qi, wqi) and np.array_equal(q, wq)
    assert np.array_equal(d_ev.to_host().reshape(-1, 4), ev)
    d_ev.free()
    params.deinit()


def test_hyperkzg_batch_open_2_16(env):
    """batchOpen of three 2^16-entry polynomials (one shorter) on a 2^16-point SRS against the oracle."""
    api, lib, ob = env
    v = 16
    gm = ob.g1_gen_multiples(1 << v)
    inf = np.zeros(1 << v, dtype=np.uint8)
    params = api.HyperKZG.SetupParams(gm, inf)
    polys = [_rand(ob, 830, 1 << v), _rand(ob, 831, 1 << v), _rand(ob, 832, 40000)]
    point = _rand(ob, 833, v)
    got = api.HyperKZG.batchOpen(params, polys, point)
    wq, wqi, wev, wfin, wgam = ob.hyperkzg_batch_open(gm, inf, polys, point)
    assert len(got["quotient_commitments"]) == wq.shape[0] == v
    for i, (q, qi) in enumerate(got["quotient_commitments"]):
        assert qi == wqi[i] and np.array_equal(q, wq[i]), i
    assert np.array_equal(got["evaluations"], wev) and np.array_equal(got["final_eval"], wfin)
    assert np.array_equal(got["batching_challenge"], wgam)
    params.deinit()


def _ptau(sections):
    out = b"ptau" + (1).to_bytes(4, "little") + len(sections).to_bytes(4, "little")
    for typ, payload in sections:
        out += typ.to_bytes(4, "little") + len(payload).to_bytes(8, "little") + payload
    return out


def test_srs_ptau_loader(env):
    """loadFromPtau's G1 side (src/poly/commitment/srs.zig:733-900) with the reference's own test shapes (:935-1000): header-only
    file -> power / ceremony power; bad magic -> InvalidFileFormat; 4 bytes -> TruncatedData; a G1 record is x | y little-endian;
    plus TauG1 / AlphaTauG1 sections of real points, an infinity record, a count capped by the power, and a point off the curve."""
    api, lib, ob = env
    hdr = (32).to_bytes(4, "little") + bytes(32) + (2).to_bytes(4, "little") + (2).to_bytes(4, "little")
    got = api.srs_g1_from_ptau(_ptau([(1, hdr)]))
    assert got["power"] == 2 and got["ceremony_power"] == 2 and got["powers_of_tau_g1"][0].shape == (0, 8)
    with pytest.raises(api.SRSError, match="InvalidFileFormat"):
        api.srs_g1_from_ptau(bytes(24))
    with pytest.raises(api.SRSError, match="TruncatedData"):
        api.srs_g1_from_ptau(b"ptau")
    with pytest.raises(api.SRSError, match="UnsupportedFormat"):
        api.srs_g1_from_ptau(b"ptau" + (2).to_bytes(4, "little") + bytes(8))
    # points: canonical little-endian coordinates of (i+1)G; record 3 is the identity
    gm = ob.g1_gen_multiples(9)
    canon = ob.f_from_mont(ob.FP, gm.reshape(-1, 4)).reshape(9, 8)
    recs = [canon[i].tobytes() for i in range(9)]
    recs[3] = bytes(64)
    tau = b"".join(recs)
    got = api.srs_g1_from_ptau(_ptau([(1, hdr), (2, tau), (4, b"".join(recs[:4])), (3, bytes(128 * 5)), (6, bytes(128))]))
    xy, inf = got["powers_of_tau_g1"]
    assert xy.shape == (7, 8)  # min(2*2^2 - 1, 9)
    want = gm[:7].copy()
    want[3] = 0
    assert np.array_equal(xy, want) and list(inf) == [0, 0, 0, 1, 0, 0, 0]
    axy, ainf = got["alpha_tau_g1"]
    assert axy.shape == (4, 8) and np.array_equal(axy[:3], gm[:3]) and ainf[3] == 1 and got["beta_tau_g1"] is None
    assert len(got["tau_g2_raw"]) == 640 and len(got["beta_g2_raw"]) == 128
    # the loaded points are a usable SRS: commit with them
    params = api.HyperKZG.SetupParams(xy, inf)
    ev = _rand(ob, 77, 7)
    c, ci = api.HyperKZG.commit(params, ev)
    w, wi = ob.hyperkzg_commit(xy, inf, ev)
    assert ci == wi and np.array_equal(c, w)
    params.deinit()
    bad = bytearray(tau)
    bad[64] ^= 1  # x of the second point
    with pytest.raises(api.SRSError, match="PointNotOnCurve"):
        api.srs_g1_from_ptau(_ptau([(1, hdr), (2, bytes(bad))]))


def test_poly_classes_and_run_sumcheck(env):
    api, lib, ob = env
    ev = _rand(ob, 20, 256)
    r = _rand(ob, 21, 1)[0]
    p = api.DensePolynomial(ev)
    assert p.num_vars == 8 and p.len() == 256
    assert np.array_equal(p.bindFirst(r).evaluations, ob.fr_bind_high(ev, r))
    p.bindLow(r)
    assert p.num_vars == 7 and np.array_equal(p.evaluations, ob.fr_bind_low(ev, r))
    pt = _rand(ob, 22, 9)
    assert np.array_equal(api.EqPolynomial(pt).evals(), ob.fr_eq_table(pt))
    sc = _rand(ob, 23, 1)[0]
    assert np.array_equal(api.EqPolynomial.evalsSliceWithScaling(pt, sc), ob.fr_eq_table(pt, sc))
    res = api.runSumcheck(api.DensePolynomial(ev))
    wc, wr, wch, wfin, wok = ob.run_sumcheck(ev)
    assert res["result"] and wok == 1
    assert np.array_equal(res["claim"], wc) and np.array_equal(np.array(res["rounds"]), wr)
    assert np.array_equal(np.array(res["final_point"]), wch) and np.array_equal(res["final_eval"], wfin)
    inter = api.runSumcheckInteractive(api.DensePolynomial(ev))  # host verifier, one round trip per round: same transcript
    assert inter["result"] and np.array_equal(inter["claim"], wc) and np.array_equal(np.array(inter["rounds"]), wr)
    assert np.array_equal(np.array(inter["final_point"]), wch) and np.array_equal(inter["final_eval"], wfin)
    with pytest.raises(AssertionError):
        api.DensePolynomial(ev[:100])  # length must be a power of two (src/poly/mod.zig:36-37)


def test_c_abi_is_reentrant(env):
    """MSM.compute is called from std.Thread workers in the reference (src/msm/mod.zig:637,732): several host
    threads hammer one handle and their own handles concurrently; every result must be the oracle's."""
    api, lib, ob = env
    n = 3000
    gm = ob.g1_gen_multiples(n)
    shared = lib.Bases.upload(gm)
    scs = [_rand(ob, 100 + t, n) for t in range(6)]
    want = [ob.msm_g1(gm, None, s) for s in scs]
    errors = []

    def worker(t):
        try:
            own = lib.Bases.upload(gm[: 1000 + 100 * t])
            for _ in range(4):
                got, inf = shared.msm(scs[t])
                if inf != want[t][1] or not np.array_equal(got, want[t][0]):
                    errors.append(("shared", t))
                g2, i2 = own.msm(scs[t][: 1000 + 100 * t])
                w2, wi2 = ob.msm_g1(gm[: 1000 + 100 * t], None, scs[t][: 1000 + 100 * t])
                if i2 != wi2 or not np.array_equal(g2, w2):
                    errors.append(("own", t))
                s = lib.SumcheckSession.open(scs[t][:1024])
                g0, g1 = s.round_sums()
                w0, w1 = ob.fr_sum_halves(scs[t][:1024])
                if not (np.array_equal(g0, w0) and np.array_equal(g1, w1)):
                    errors.append(("sums", t))
                s.close()
            own.free()
        except Exception as e:  # noqa: BLE001
            errors.append((repr(e), t))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    shared.free()
    assert not errors, errors


def test_srs_raw_binary_g1_round_trip(env):
    """SRS raw binary wire format, G1 section (src/poly/commitment/srs.zig:256-306,358-408): bytes written from
    the oracle's mock SRS parse back to the same Montgomery limbs, commit through them matches, infinity is all-zero."""
    api, lib, ob = env
    from oracle import pymodel as pm
    n = 33
    srs, inf = ob.hyperkzg_setup(n)
    pts = pm.mock_srs(n)
    want_bytes = n.to_bytes(4, "little") + b"".join(pm.commitment_bytes(p) for p in pts) + bytes(320)
    assert api.srs_g1_to_raw(srs, inf) == want_bytes
    xy, pinf, trailer = api.srs_g1_from_raw(want_bytes)
    assert np.array_equal(xy, srs) and not pinf.any() and trailer == bytes(320)
    blob = bytearray(want_bytes)
    blob[4 + 64 * 5: 4 + 64 * 6] = bytes(64)  # point 5 := infinity
    xy2, pinf2, _ = api.srs_g1_from_raw(bytes(blob))
    assert pinf2[5] == 1 and pinf2.sum() == 1 and not xy2[5].any()
    ev = _rand(ob, 60, n)
    b = lib.Bases.upload(xy2, pinf2)
    got, ginf = b.msm(ev)
    b.free()
    winf = inf.copy()
    winf[5] = 1
    want, wi = ob.msm_g1(srs, winf, ev)
    assert ginf == wi and np.array_equal(got, want)
    with pytest.raises(api.SRSError):
        api.srs_g1_from_raw(want_bytes[:100])
    bad = bytearray(want_bytes)
    bad[4 + 64 * 3 + 63] ^= 1  # y of point 3 off the curve -> PointNotOnCurve (srs.zig:93-96)
    with pytest.raises(api.SRSError):
        api.srs_g1_from_raw(bytes(bad))
    oc = lib.g1_is_on_curve_batch(np.concatenate([srs, srs[:1] ^ np.uint64(2)]), None)
    assert oc[:n].all() and oc[n] == 0


def test_dense_evaluate(env):
    """DensePolynomial.evaluate (src/poly/mod.zig:73-92): boolean corners select entries (commitment/mod.zig:1448-1472),
    random points match the oracle's term-by-term expansion; open()'s final evaluation is evaluate(reversed point)."""
    api, lib, ob = env
    for v in (0, 1, 3, 6, 10):
        ev = _rand(ob, 70 + v, 1 << v)
        p = api.DensePolynomial(ev)
        pt = _rand(ob, 80 + v, v)
        assert np.array_equal(p.evaluate(pt), ob.fr_dense_evaluate(ev, pt)), v
    ev = _rand(ob, 90, 8)
    for idx in range(8):
        assert np.array_equal(api.DensePolynomial(ev).evaluate(U.fr([(idx >> j) & 1 for j in range(3)])), ev[idx])
