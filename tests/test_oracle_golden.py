"""Pins the CPU oracle (oracle/zolt_oracle.c) against the reference's own golden data
and against the independent Python big-int model. CPU-only (-m "not gpu")."""
import json
import os

import numpy as np
import pytest

from oracle import binding as ob
from oracle import pymodel as pm
from tests import util as U

FR, FP = ob.FR, ob.FP


# ---------------------------------------------------------------- reference fixtures
def test_bytecode_commitment_matches_reference_proof(golden_dir):
    """logs/zolt_proof_regular.bin bytes 8..72 = HyperKZG.commit(bytecode poly) of
    examples/fibonacci.elf (src/zkvm/mod.zig:1519-1538 -> commitment/mod.zig:239-255
    -> msm/mod.zig:355-438), serialised by commitment_types.zig:49-54."""
    elf = open(os.path.join(golden_dir, "fibonacci.elf"), "rb").read()
    proof = open(os.path.join(golden_dir, "zolt_proof_regular.bin"), "rb").read()
    assert proof[:4] == b"ZOLT"
    code = elf[0x1000:0x1000 + 104]
    srs, inf = ob.hyperkzg_setup(128)
    ev = np.zeros(128, dtype=np.uint64)
    ev[:104] = np.frombuffer(code, dtype=np.uint8)
    c, ci = ob.hyperkzg_commit(srs, inf, ob.f_from_u64(FR, ev))
    assert ci == 0
    assert ob.commitment_to_bytes(c) == proof[8:72]
    # the memory commitment in the same proof is the identity -> 64 zero bytes
    z, zi = ob.hyperkzg_commit(srs, inf, np.zeros((0, 4), dtype=np.uint64))
    assert zi == 1 and ob.commitment_to_bytes(z) == proof[232:296] == bytes(64)
    # independent model agrees too
    r = pm.msm(pm.mock_srs(128), [int(x) for x in ev])
    assert pm.commitment_bytes(r) == proof[8:72]


def test_register_commitment_matches_reference_proof(golden_dir):
    """logs/zolt_proof_regular.bin bytes 488..552 = HyperKZG.commit of the register polynomial poly[i] = rd_value of trace step i
    (src/zkvm/mod.zig:1585-1617), 54 executed steps of examples/fibonacci.elf padded to 256 (logs/zolt.log:23-27): a second
    commitment the reference itself produced, regenerated from the ELF through a minimal RV64 interpreter (tests/util.py)."""
    elf = open(os.path.join(golden_dir, "fibonacci.elf"), "rb").read()
    proof = open(os.path.join(golden_dir, "zolt_proof_regular.bin"), "rb").read()
    vals = U.fibonacci_rd_values(elf)
    assert len(vals) == 54 and vals[:4] == [0x8000, 0x8001, 0x80010000, 0x80000010] and vals[-1] == 0x80000014
    ev = np.zeros(256, dtype=np.uint64)
    ev[:54] = np.array(vals, dtype=np.uint64)
    srs, inf = ob.hyperkzg_setup(256)
    c, ci = ob.hyperkzg_commit(srs, inf, ob.f_from_u64(FR, ev))
    assert ci == 0 and ob.commitment_to_bytes(c) == proof[488:552]
    r = pm.msm(pm.mock_srs(256), [int(x) for x in ev])
    assert pm.commitment_bytes(r) == proof[488:552]


def test_proof_header_bytes_match_reference_proof(golden_dir):
    """The first 744 bytes of the reference's captured proof (ZOLT v1 header: magic, version, the bytecode / memory / register
    proofs' twelve commitment slots and the legacy field element, src/zkvm/serialization.zig:283-306) re-serialised by the host
    mirror from commitments computed by the oracle: byte-identical."""
    from zolt_amd import api
    elf = open(os.path.join(golden_dir, "fibonacci.elf"), "rb").read()
    proof = open(os.path.join(golden_dir, "zolt_proof_regular.bin"), "rb").read()
    srs, inf = ob.hyperkzg_setup(256)
    bc = np.zeros(128, dtype=np.uint64)
    bc[:104] = np.frombuffer(elf[0x1000:0x1000 + 104], dtype=np.uint8)
    reg = np.zeros(256, dtype=np.uint64)
    reg[:54] = np.array(U.fibonacci_rd_values(elf), dtype=np.uint64)
    hdr = api.serialize_zolt_proof_header({
        "bytecode.commitment": ob.hyperkzg_commit(srs, inf, ob.f_from_u64(FR, bc)),
        "memory.commitment": ob.hyperkzg_commit(srs, inf, np.zeros((0, 4), dtype=np.uint64)),
        "register.commitment": ob.hyperkzg_commit(srs, inf, ob.f_from_u64(FR, reg))})
    assert hdr == proof[:744]
    assert api.parse_zolt_proof_commitments(hdr + bytes(64))["register.commitment"] == proof[488:552]


def test_field_constants_and_kats():
    """src/field/mod.zig:16-75 constants; KATs :1101-1140 (3*7=21, 7*7^-1=1, 2^3=8)."""
    for f, mod in ((FR, pm.R_MOD), (FP, pm.P_MOD)):
        one = ob.f_from_u64(f, np.array([1], dtype=np.uint64))[0]
        assert pm.from_limbs(one) == (1 << 256) % mod
        a = ob.f_from_u64(f, np.array([3, 7, 2], dtype=np.uint64))
        assert U.fr_to_int(ob.f_mul(f, a[0], a[1])) == 21 if f == FR else True
        prod = ob.f_mul(f, a[1], ob.f_inv(f, a[1]))
        assert np.array_equal(prod, one)
        two = a[2]
        assert np.array_equal(ob.f_mul(f, ob.f_mul(f, two, two), two), ob.f_from_u64(f, np.array([8], dtype=np.uint64))[0])
        z = np.zeros(4, dtype=np.uint64)
        assert np.array_equal(ob.f_neg(f, z), z)  # neg(0) = 0
        assert np.array_equal(ob.f_inv(f, z), z)  # inverse(0) = null -> reported as 0


@pytest.mark.parametrize("f,mod", [(FR, pm.R_MOD), (FP, pm.P_MOD)])
def test_field_random_vs_bigint(f, mod):
    n = 2000
    a_raw = U.random_raw256(11 + f, n)
    b_raw = U.random_raw256(23 + f, n)
    # edge values, incl. values >= modulus for the fromBytes path (src/field/mod.zig:171-184)
    edges = [0, 1, mod - 1, mod, mod + 1, (1 << 256) - 1, (1 << 256) % mod]
    for i, e in enumerate(edges):
        a_raw[i] = pm.limbs(e)
        b_raw[-1 - i] = pm.limbs(e)
    a = ob.f_to_mont(f, a_raw)
    b = ob.f_to_mont(f, b_raw)
    ai = [pm.from_limbs(x) % mod for x in a_raw]
    bi = [pm.from_limbs(x) % mod for x in b_raw]
    for name, fn, py in (("mul", ob.f_mul, lambda x, y: x * y), ("add", ob.f_add, lambda x, y: x + y),
                         ("sub", ob.f_sub, lambda x, y: x - y)):
        got = fn(f, a, b)
        for i in range(n):
            assert pm.from_mont(pm.from_limbs(got[i]), mod) == py(ai[i], bi[i]) % mod, name
            assert pm.from_limbs(got[i]) < mod
    back = ob.f_from_mont(f, a)
    for i in range(n):
        assert pm.from_limbs(back[i]) == ai[i]
    inv = ob.f_inv(f, a[:64])
    for i in range(64):
        if ai[i]:
            assert pm.from_mont(pm.from_limbs(inv[i]), mod) == pow(ai[i], -1, mod)


def test_window_kats():
    """src/msm/mod.zig:827-851."""
    for n, c in ((1, 1), (7, 1), (8, 2), (31, 2), (32, 3), (127, 3), (128, 4), (511, 4), (512, 5), (2047, 5),
                 (2048, 6), (8191, 6), (8192, 7), (32767, 7), (32768, 8), (1 << 20, 8)):
        assert ob.optimal_window_size(n) == c
    s = U.fr([202])[0]  # 0b11001010
    assert ob.get_window(s, 0, 4) == 10 and ob.get_window(s, 1, 4) == 12
    # limb-crossing window (c=7, bit 63)
    v = (0x55 << 60) | 0x123
    s = U.fr([v])[0]
    for w in range(37):
        assert ob.get_window(s, w, 7) == (v >> (7 * w)) & 0x7F


def _xy_inf(case):
    pts = [(int(p[0], 16), int(p[1], 16)) for p in case["points"]]
    return U.points_xy(pts), np.array(case["inf"], dtype=np.uint8), U.fr_hex(case["scalars"])


def test_msm_golden_vectors():
    vec = U.load_vectors()
    for case in vec["msm"]:
        xy, inf, sc = _xy_inf(case)
        out, oinf = ob.msm_g1(xy, inf, sc)
        got = U.point_from_xy(out, oinf)
        want = None if case["result"] is None else tuple(int(h, 16) for h in case["result"])
        assert got == want, case["n"]
        if case["n"] >= 4096:  # ParallelMSM only splits at >= 1024 points per thread
            continue
        out2, oinf2 = ob.msm_g1_parallel(xy, inf, sc, 4)
        assert U.point_from_xy(out2, oinf2) == want


def test_msm_reference_edge_semantics():
    """src/msm/mod.zig:802-825,875-891,938-966 — 0*P = inf, 1*P = P, zero scalars, empty."""
    g = U.points_xy([pm.G1])
    out, inf = ob.g1_scalar_mul(g[0], 0, U.fr([0])[0])
    assert inf == 1
    out, inf = ob.g1_scalar_mul(g[0], 0, U.fr([1])[0])
    assert inf == 0 and np.array_equal(out, g[0])
    pts = ob.g1_gen_multiples(10)
    out, inf = ob.msm_g1(pts, None, np.zeros((10, 4), dtype=np.uint64))
    assert inf == 1 and not out.any()
    out, inf = ob.msm_g1(pts[:0], None, np.zeros((0, 4), dtype=np.uint64))
    assert inf == 1
    out, inf = ob.msm_g1_parallel(pts[:8], None, np.zeros((8, 4), dtype=np.uint64), 4)
    assert inf == 1
    # scalarMul(P,2) == double (src/integration_tests.zig:145-161) and affine double == Jacobian double
    d1, i1 = ob.g1_scalar_mul(g[0], 0, U.fr([2])[0])
    d2, i2 = ob.g1_double_affine(g[0], 0)
    assert i1 == i2 == 0 and np.array_equal(d1, d2)
    assert U.point_from_xy(d1, 0) == pm.ec_add(pm.G1, pm.G1)


def test_msm_bench_family_closed_form():
    """Bases (i+1)G, scalars 7i+13 (src/bench.zig:261-268) against the closed form."""
    vec = U.load_vectors()
    gm = ob.g1_gen_multiples(1000)
    for k, want in enumerate(vec["generator_multiples"]):
        assert U.point_from_xy(gm[k], 0) == tuple(int(h, 16) for h in want)
    for case in vec["msm_bench_family"]:
        n = case["n"]
        sc = ob.f_from_u64(FR, np.array([7 * i + 13 for i in range(n)], dtype=np.uint64))
        out, inf = ob.msm_g1(gm[:n], None, sc)
        assert U.point_from_xy(out, inf) == tuple(int(h, 16) for h in case["result"])


def test_pippenger_window_paths_vs_closed_form():
    """Exercise every optimalWindowSize branch incl. c=7 limb-crossing windows and c=8."""
    gm = ob.g1_gen_multiples(40000)
    for n in (8, 40, 200, 600, 2500, 9000, 33000, 40000):
        raw = U.random_raw256(n, n)
        sc = ob.f_to_mont(FR, raw)
        ints = [pm.from_limbs(x) % pm.R_MOD for x in raw]
        want = pm.msm_generator_multiples(range(1, n + 1), ints)
        out, inf = ob.msm_g1(gm[:n], None, sc)
        assert U.point_from_xy(out, inf) == want, n
    out2, inf2 = ob.msm_g1_parallel(gm[:40000], None, sc, 8)
    assert U.point_from_xy(out2, inf2) == want


def test_mock_srs_and_batch_commit():
    vec = U.load_vectors()
    srs, inf = ob.hyperkzg_setup(6)
    for i, want in enumerate(vec["mock_srs"]):
        assert U.point_from_xy(srs[i], inf[i]) == tuple(int(h, 16) for h in want)
        assert ob.g1_is_on_curve(srs[i])
    # batchCommit[i] == commit(poly_i) (src/poly/commitment/mod.zig:1392-1420)
    srs, inf = ob.hyperkzg_setup(16)
    polys = [ob.f_to_mont(FR, U.random_raw256(5 + k, 16)) for k in range(3)]
    outs, oinf = ob.msm_g1_batch(srs, inf, polys)
    for k in range(3):
        c, ci = ob.hyperkzg_commit(srs, inf, polys[k])
        assert np.array_equal(c, outs[k]) and ci == oinf[k]


# ---------------------------------------------------------------- poly / sumcheck
def test_bind_low_kats():
    """src/poly/mod.zig:816-888."""
    got = ob.fr_bind_low(U.fr([1, 2, 3, 4]), U.fr([3])[0])
    assert [U.fr_to_int(x) for x in got] == [4, 6]
    got = ob.fr_bind_low(U.fr([10, 20, 30, 40, 50, 60, 70, 80]), U.fr([5])[0])
    assert [U.fr_to_int(x) for x in got] == [60, 80, 100, 120]


def test_sumcheck_kats():
    """src/subprotocols/mod.zig:366-461."""
    t = U.fr([1, 2, 3, 4])
    g0, g1 = ob.fr_sum_halves(t)
    assert (U.fr_to_int(g0), U.fr_to_int(g1)) == (3, 7)
    t2 = ob.fr_bind_high(t, U.fr([2])[0])
    assert [U.fr_to_int(x) for x in t2] == [5, 6]
    claim, rounds, chals, fin, ok = ob.run_sumcheck(U.fr(range(1, 9)))
    assert U.fr_to_int(claim) == 36 and ok == 1


def test_eq_table_golden_and_properties():
    vec = U.load_vectors()
    for case in vec["eq_table"]:
        r = U.fr_hex(case["r"]) if case["r"] else np.zeros((0, 4), dtype=np.uint64)
        got = ob.fr_eq_table(r)
        assert [U.fr_to_int(x) for x in got] == [int(h, 16) for h in case["table"]]
        if len(case["r"]):
            assert np.array_equal(ob.fr_eq_table_append_lsb(r), got)  # split_eq build order, same values
        assert sum(U.fr_to_int(x) for x in got) % pm.R_MOD == 1  # partition of unity (poly/mod.zig:689-753)
    # scaling factor
    r = U.fr_hex(vec["eq_table"][3]["r"])
    sc = U.fr([12345])[0]
    got = ob.fr_eq_table(r, sc)
    assert [U.fr_to_int(x) for x in got] == [int(h, 16) * 12345 % pm.R_MOD for h in vec["eq_table"][3]["table"]]
    # the captured run's 13 tau challenges (logs/zolt.log:47-59) as a bigger input
    taus = json.load(open(os.path.join(U.GOLDEN, "stage1_tau.json")))["tau_hex"]
    tv = [int(h, 16) % pm.R_MOD for h in taus]
    got = ob.fr_eq_table(U.fr(tv))
    want = pm.eq_table(tv)
    assert [U.fr_to_int(x) for x in got[:64]] == want[:64] and U.fr_to_int(got[-1]) == want[-1]


def test_folds_and_sumcheck_golden():
    vec = U.load_vectors()
    for case in vec["folds"]:
        t, r = U.fr_hex(case["table"]), U.fr_hex([case["r"]])[0]
        assert [U.fr_to_int(x) for x in ob.fr_bind_high(t, r)] == [int(h, 16) for h in case["bind_high"]]
        assert [U.fr_to_int(x) for x in ob.fr_bind_low(t, r)] == [int(h, 16) for h in case["bind_low"]]
        assert [U.fr_to_int(x) for x in ob.fr_bind_low_2mul(t, r)] == [int(h, 16) for h in case["bind_low"]]
    for case in vec["sumcheck"]:
        claim, rounds, chals, fin, ok = ob.run_sumcheck(U.fr_hex(case["evals"]))
        assert U.fr_to_int(claim) == int(case["claim"], 16)
        assert [[U.fr_to_int(c) for c in rd] for rd in rounds] == [[int(h, 16) for h in rd] for rd in case["rounds"]]
        assert [U.fr_to_int(c) for c in chals] == [int(h, 16) for h in case["challenges"]]
        assert U.fr_to_int(fin) == int(case["final_eval"], 16) and ok == int(case["ok"])


def test_dense_evaluate_and_hyperkzg_open_fold():
    """DensePolynomial.evaluate corners (commitment/mod.zig:1448-1472) and open()'s final eval."""
    t = ob.f_to_mont(FR, U.random_raw256(77, 8))
    for idx in range(8):
        pt = U.fr([(idx >> j) & 1 for j in range(3)])
        assert np.array_equal(ob.fr_dense_evaluate(t, pt), t[idx])
    srs, inf = ob.hyperkzg_setup(8)
    point = ob.f_to_mont(FR, U.random_raw256(78, 3))
    q, qinf, fin = ob.hyperkzg_open(srs, inf, t, point, np.zeros(4, dtype=np.uint64))
    # open() folds the HIGH half first: final = evaluate with point reversed in LSB-first order
    assert np.array_equal(fin, ob.fr_dense_evaluate(t, point[::-1].copy()))
    cur = t
    for i in range(3):
        half = len(cur) // 2
        c, ci = ob.hyperkzg_commit(srs, inf, ob.f_sub(FR, cur[half:], cur[:half]))
        assert np.array_equal(c, q[i]) and ci == qinf[i]
        cur = ob.fr_bind_high(cur, point[i])


def _gruen_fixture(golden_dir):
    d = json.load(open(os.path.join(golden_dir, "stage4_gruen_eq.json")))
    tau = np.array([[int(x) for x in row] for row in d["r_cycle_be_mont_limbs"]], dtype=np.uint64)  # raw Montgomery limbs
    return d, tau


def test_gruen_split_eq_tables_match_reference_log(golden_dir):
    """The only eq-table values the reference itself produced and kept: logs/zolt.log [STAGE4_GRUEN_INIT]
    (stage4_gruen_prover.zig:266-312): E_out = eq table over w_out = tau[0..m], E_in over w_in = tau[m..n-1]
    (src/poly/split_eq.zig:91-171), entries printed as canonical little-endian bytes (F.toBytes, field/mod.zig:685-695).
    Pins EqPolynomial.evalsSliceWithScaling (A17) and the split-eq append-LSB build (A19) of the oracle AND of the
    independent big-int model against reference-produced numbers."""
    d, tau = _gruen_fixture(golden_dir)
    n, m = d["n"], d["m"]
    can = ob.f_from_mont(FR, tau)
    assert can[n - 1].tobytes().hex() == d["current_w_canonical_le_hex"]  # decoding of the Montgomery limbs is right
    assert ob.f_from_mont(FR, ob.f_from_u64(FR, np.array([1], dtype=np.uint64)))[0].tobytes().hex() == d["current_scalar_canonical_le_hex"]
    for name, sl, ln in (("E_out", slice(0, m), d["E_out_len"]), ("E_in", slice(m, n - 1), d["E_in_len"])):
        want = d[name + "_first4_canonical_le_hex"]
        for build in (ob.fr_eq_table, ob.fr_eq_table_append_lsb):
            t = build(tau[sl])
            assert len(t) == ln
            tc = ob.f_from_mont(FR, t)
            assert [tc[i].tobytes().hex() for i in range(4)] == want, (name, build.__name__)
        # independent model: canonical ints in, canonical ints out
        r_int = [pm.from_mont(pm.from_limbs(x), pm.R_MOD) for x in tau[sl]]
        tp = pm.eq_table(r_int)
        assert len(tp) == ln
        assert [int(v).to_bytes(32, "little").hex() for v in tp[:4]] == want, name


def test_gruen_split_eq_state_machine_reference_inline_tests(golden_dir):
    """src/poly/split_eq.zig:525-733 — the reference's six GruenSplitEqPolynomial tests restated against the oracle's
    restatement of the struct (oracle.binding.GruenSplitEq), plus the prefix tables of the captured run (fixture)."""
    F = lambda v: ob.f_from_u64(FR, np.array([v], dtype=np.uint64))[0]
    one = F(1)
    sub = lambda a, b: ob.f_sub(FR, a[None, :], b[None, :])[0]
    mul = lambda a, b: ob.f_mul(FR, a[None, :], b[None, :])[0]
    add = lambda a, b: ob.f_add(FR, a[None, :], b[None, :])[0]
    tau = np.stack([F(2), F(3), F(5)])
    p = ob.GruenSplitEq(tau)
    assert p.current_index == 3 and np.array_equal(p.current_scalar, one)
    assert (p.num_x_in, p.num_x_out, len(p.E_in_vec), len(p.E_out_vec)) == (1, 1, 2, 2)
    assert np.array_equal(p.E_out_vec[1], np.stack([sub(one, F(2)), F(2)]))
    assert np.array_equal(p.E_in_vec[1], np.stack([sub(one, F(3)), F(3)]))
    p = ob.GruenSplitEq(np.stack([F(2), F(3)]))
    p.bind(F(5))
    assert p.current_index == 1 and np.array_equal(p.current_scalar, F(23))
    p = ob.GruenSplitEq(np.stack([F(1), F(2)]))
    rp = p.computeCubicRoundPoly(F(10), F(3), F(100))
    assert np.array_equal(add(rp[0], rp[1]), F(100))
    p = ob.GruenSplitEq(np.stack([F(3), F(5), F(7), F(11)]))
    t = p.getFullEqTable()
    m3, m5, m7, m11 = (sub(one, F(v)) for v in (3, 5, 7, 11))
    assert len(t) == 16
    assert np.array_equal(t[0], mul(mul(m3, m5), mul(m7, m11))) and np.array_equal(t[15], F(3 * 5 * 7 * 11))
    assert np.array_equal(t[5], mul(mul(m3, F(5)), mul(m7, F(11)))) and np.array_equal(t[10], mul(mul(F(3), m5), mul(F(7), m11)))
    assert np.array_equal(p.getEActiveForWindow(1), one[None, :])
    assert np.array_equal(p.getEActiveForWindow(2), np.stack([m7, F(7)]))
    assert np.array_equal(p.getEActiveForWindow(3), np.stack([mul(m5, m7), mul(m5, F(7)), mul(F(5), m7), F(35)]))
    # every prefix table is the eq table of the prefix (A17 build), and the captured run's E_out / E_in are the last ones
    d, tau = _gruen_fixture(golden_dir)
    n, m = d["n"], d["m"]
    g = ob.GruenSplitEq(tau)
    assert len(g.E_out_vec) == m + 1 and len(g.E_in_vec) == n - 1 - m + 1
    for k, tab in enumerate(g.E_out_vec):
        assert np.array_equal(tab, ob.fr_eq_table(tau[:k]))
    for k, tab in enumerate(g.E_in_vec):
        assert np.array_equal(tab, ob.fr_eq_table(tau[m:m + k]))
    for name, tab in (("E_out", g.E_out_vec[-1]), ("E_in", g.E_in_vec[-1])):
        tc = ob.f_from_mont(FR, tab)
        assert len(tab) == d[name + "_len"] and [tc[i].tobytes().hex() for i in range(4)] == d[name + "_first4_canonical_le_hex"]
    # bind pops E_in first, then E_out, never table 0 (:213-248); the window tables shrink with it (:312-343)
    sizes = []
    for r in range(n):
        e_out, e_in, hib = g.getWindowEqTables(1)
        sizes.append((len(e_out), len(e_in), hib))
        assert len(e_out).bit_length() - 1 + len(e_in).bit_length() - 1 + 1 == g.current_index
        g.bind(F(1000 + r))
    assert sizes[0] == (1 << m, 1 << (n - 1 - m), n - 1 - m) and sizes[-1] == (1, 1, 0)
    assert len(g.E_out_vec) == 1 and len(g.E_in_vec) == 1 and g.current_index == 0


def test_lasso_prover_oracle_reference_inline_tests_and_bigint_model():
    """src/zkvm/lasso/prover.zig:553-688 — "basic", "rounds" and "claim tracking" restated against the oracle's LassoProver
    (the claim after every challenge is p(r) of the round polynomial just sent), then the whole protocol against an independent
    big-int model (eq by the product formula, sums and folds with plain ints)."""
    F = lambda v: ob.f_from_u64(FR, np.array([v], dtype=np.uint64))[0]
    idx = np.array([[0, 0], [1, 0], [2, 0], [3, 0]], dtype=np.uint64)
    p = ob.LassoProver(idx, 2, 3, np.stack([F(2), F(3)]))
    assert p.round == 0 and p.isAddressPhase() and not p.isComplete()
    uni = p.computeRoundPolynomial()
    assert len(uni) > 0
    p.receiveChallenge(F(7))
    assert p.round == 1
    p = ob.LassoProver(idx, 2, 3, np.stack([F(2), F(3)]))
    to_int = lambda a: pm.from_mont(pm.from_limbs(a), pm.R_MOD)
    for rnd in range(5):
        claim = to_int(p.current_claim)
        c0, c1, c2 = (to_int(c) for c in p.computeRoundPolynomial())
        assert (c0 + (c0 + c1 + c2)) % pm.R_MOD == claim
        ch = rnd + 10
        p.receiveChallenge(F(ch))
        assert to_int(p.current_claim) == (c0 + c1 * ch + c2 * ch * ch) % pm.R_MOD
    assert p.isComplete()
    # independent model, ragged cycle count, indices above 64 bits
    rng = np.random.default_rng(5)
    log_T, log_K, n = 5, 70, 21
    P = pm.R_MOD
    w = [int(rng.integers(1, 1 << 62)) for _ in range(log_T)]
    lk = [(int(rng.integers(0, 1 << 62)) << 40) ^ int(rng.integers(0, 1 << 62)) for _ in range(n)]
    lk = [x & ((1 << log_K) - 1) for x in lk]
    wm = ob.f_to_mont(FR, np.array([[x, 0, 0, 0] for x in w], dtype=np.uint64))
    idxa = np.array([[x & (2**64 - 1), x >> 64] for x in lk], dtype=np.uint64)
    lp = ob.LassoProver(idxa, log_T, log_K, wm)
    outer, inner = log_T // 2, log_T - log_T // 2

    def eq(j):
        o, i, v = j >> inner, j & ((1 << inner) - 1), 1
        for b in range(outer):
            v = v * (w[b] if (o >> b) & 1 else 1 - w[b]) % P
        for b in range(inner):
            v = v * (w[outer + b] if (i >> b) & 1 else 1 - w[outer + b]) % P
        return v

    ev = [eq(j) if j < n else 0 for j in range(1 << log_T)]
    assert [to_int(x) for x in lp.eq_evals] == ev and to_int(lp.current_claim) == sum(ev) % P
    chal = [int(rng.integers(1, 1 << 62)) for _ in range(log_T + log_K)]
    cm = ob.f_to_mont(FR, np.array([[x, 0, 0, 0] for x in chal], dtype=np.uint64))
    ln = len(ev)
    for r in range(log_T + log_K):
        c = [to_int(x) for x in lp.computeRoundPolynomial()]
        if r < log_K:
            s0 = sum(ev[j] for j in range(n) if not (lk[j] >> r) & 1) % P
            s1 = sum(ev[j] for j in range(n) if (lk[j] >> r) & 1) % P
        else:
            s0, s1 = sum(ev[:ln // 2]) % P, sum(ev[ln // 2:ln]) % P
        assert c == [s0, (s1 - s0) % P, 0], r
        assert (s0 + s1) % P == to_int(lp.current_claim)
        lp.receiveChallenge(cm[r])
        ch = chal[r]
        if r < log_K:
            for j in range(n):
                ev[j] = ev[j] * (ch if (lk[j] >> r) & 1 else 1 - ch) % P
        else:
            ev[:ln // 2] = [((1 - ch) * ev[j] + ch * ev[j + ln // 2]) % P for j in range(ln // 2)]
            ln //= 2
    assert lp.isComplete() and to_int(lp.current_claim) == ev[0]
    fin = 1
    for ch in chal[:log_K]:
        fin = fin * (1 - ch) % P
    assert to_int(lp.getFinalEval()) == fin


def test_product_form_prover_oracles_are_sound_sumchecks():
    """The oracle's restatements of the product-form prover loops (ValEvaluation / ValFinal val_evaluation.zig:554-660, OutputCheck
    output_check.zig:375-499, InstructionLookups claim reduction instruction_lookups.zig:146-270, ProductVirtualRemainder
    product_remainder.zig:269-394) run as sumchecks with a consistent initial claim (a big-int sum over the hypercube): every round
    satisfies s(0) + s(1) = claim, and the final claim equals the expression of the tables' final values — so the restated
    formulas (extrapolations, Lagrange / Vandermonde steps, Gruen's cubic) are pinned by the protocol's own algebra."""
    P = pm.R_MOD
    rng = np.random.default_rng(11)
    to_int = lambda a: pm.from_mont(pm.from_limbs(a), P)
    to_mont = lambda v: ob.f_to_mont(FR, np.array([[(v >> (64 * i)) & (2**64 - 1) for i in range(4)]], dtype=np.uint64))[0]
    rnd = lambda n: ob.f_to_mont(FR, rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64))
    v, n = 5, 32
    for three in (True, False):
        inc, wa, lt = rnd(n), rnd(n), (rnd(n) if three else None)
        cols = [[to_int(x) for x in t] for t in ((inc, wa, lt) if three else (inc, wa))]
        claim = 0
        for row in zip(*cols):
            t = 1
            for x in row:
                t = t * x % P
            claim = (claim + t) % P
        p = ob.ValEvaluationProver(inc, wa, lt, to_mont(claim))
        for _ in range(v):
            ev = p.computeRoundPolynomial()
            assert (to_int(ev[0]) + to_int(ev[1])) % P == to_int(p.current_claim)
            p.bindChallengeWithPoly(rnd(1)[0], ev)
        fin = 1
        for x in p.getFinalClaims():
            fin = fin * to_int(x) % P
        assert fin == to_int(p.current_claim)
    tabs = [rnd(n) for _ in range(5)]
    I = [[to_int(x) for x in t] for t in tabs]
    claim = sum(I[0][j] * I[1][j] * (I[2][j] - I[3][j]) for j in range(n)) % P
    p = ob.OutputSumcheckProver(*tabs, to_mont(claim))
    for _ in range(v):
        ev = p.roundEvals()
        assert (to_int(ev[0]) + to_int(ev[1])) % P == to_int(p.current_claim)
        c = ob.interpolate_degree3(ev)  # the Vandermonde inverse reproduces the evaluations
        ci = [to_int(x) for x in c]
        assert [sum(ci[k] * t**k for k in range(4)) % P for t in range(4)] == [to_int(x) for x in ev]
        assert np.array_equal(p.computeRoundPolynomial(), np.stack([c[0], c[2], c[3]]))
        ch = rnd(1)[0]
        p.bindChallenge(ch)
        p.updateClaim(ev, ch)
        assert np.array_equal(p.current_claim, ob.raf_update_claim(ev, ch))  # both claim-update forms agree
    f = p.getFinalClaims()
    assert to_int(f["eq_r_address"]) * to_int(f["io_mask"]) * (to_int(f["val_final"]) - to_int(f["val_io"])) % P == to_int(p.current_claim)
    tabs = [rnd(n) for _ in range(4)]
    I = [[to_int(x) for x in t] for t in tabs]
    gamma = rnd(1)[0]
    g = to_int(gamma)
    claim = sum(I[0][j] * (I[1][j] + g * I[2][j] + g * g * I[3][j]) for j in range(n)) % P
    p = ob.InstructionLookupsClaimReduction(*tabs, gamma, to_mont(claim))
    for _ in range(v):
        ev = p.computeRoundPolynomialCubic()
        ch = rnd(1)[0]
        p.bindChallenge(ch)
        p.updateClaim(ev, ch)
    f = p.getOpeningClaims()
    assert to_int(p.t[0][0]) * (to_int(f["lookup_output"]) + g * to_int(f["left_operand"]) + g * g * to_int(f["right_operand"])) % P == to_int(p.current_claim)
    left, right, tau, kernel = rnd(n), rnd(n), rnd(v), rnd(1)[0]
    eq = ob.fr_eq_table(tau, kernel)
    claim = sum(to_int(a) * to_int(b) * to_int(c) for a, b, c in zip(left, right, eq)) % P
    p = ob.ProductRemainderProver(left, right, tau, kernel, to_mont(claim))
    for _ in range(v):
        ev = p.roundEvals()
        assert (to_int(ev[0]) + to_int(ev[1])) % P == to_int(p.current_claim)
        ch = rnd(1)[0]
        p.bindChallenge(ch)
        p.updateClaim(ev, ch)
    assert to_int(p.getFinalClaim()) * to_int(p.split_eq.current_scalar) % P == to_int(p.current_claim)


def test_eq_plus_one_reference_inline_test_and_shift_identity():
    """src/poly/mod.zig:890-943 "EqPlusOnePolynomial basic" restated against the oracle's mle AND the host mirror's; then the fact the
    device table rests on: over the cube eq+1(r, j) = eq(r, j - 1), eq+1(r, 0) = 0 — checked with the oracle evaluating the general
    formula at every cube point, the way computeEqPlusOneEvals does (:530-548)."""
    from zolt_amd import api
    F = lambda v: ob.f_from_u64(FR, np.array([v], dtype=np.uint64))[0]
    zero, one = F(0), F(1)
    cases = [((zero, zero), (zero, one), one), ((zero, one), (one, zero), one), ((one, zero), (one, one), one),
             ((one, one), (zero, zero), zero), ((zero, zero), (one, zero), zero)]
    for x, y, want in cases:
        assert np.array_equal(ob.eq_plus_one_mle(np.stack(x), np.stack(y)), want)
        assert np.array_equal(api.EqPlusOnePolynomial.mle(np.stack(x), np.stack(y)), want)
    rng = np.random.default_rng(17)
    for v in range(0, 8):
        r = ob.f_to_mont(FR, rng.integers(0, 1 << 62, size=(v, 4), dtype=np.uint64))
        t, eq = ob.eq_plus_one_table(r), ob.fr_eq_table(r)
        assert not t[0].any() and np.array_equal(t[1:], eq[:-1]), v
        if v:
            y = ob.f_to_mont(FR, rng.integers(0, 1 << 62, size=(v, 4), dtype=np.uint64))  # a non-boolean point: the two mle restatements agree
            assert np.array_equal(ob.eq_plus_one_mle(r, y), api.EqPlusOnePolynomial(r).evaluate(y))


def test_round_polynomials_from_the_unfolded_tables():
    """Round-2 review, "strengthen the algebra-only pins": s(0) + s(1) = claim cannot see a restatement that folds the wrong pair
    order consistently on both sides. Here every round polynomial of the product-form restatements is computed a second time from the
    ORIGINAL tables by the definition of the sumcheck it belongs to — no folded state, no shared code:

        s_k(t) = sum over x in {0,1}^(v-k-1) of  F( T_j~(r_0, ..., r_(k-1), t, x) for the tables j ),
        T~(z_0, ..., z_k, x) = sum over b in {0,1}^(k+1) of  prod_i (b_i ? z_i : 1 - z_i) * T[b_0 + 2 b_1 + ... + 2^k b_k + 2^(k+1) x]

    — variable i of the protocol IS bit i of the table index (LowToHigh: the pair (2i, 2i + 1) is bound first, val_evaluation.zig:609-628,
    output_check.zig:449-480, instruction_lookups.zig:240-270, product_remainder.zig:357-394) — with exact integers, at t = 0..3, for
    ValEvaluation / ValFinal (product of the tables), OutputCheck (eq * io * (vf - vio)), the InstructionLookups claim reduction
    (eq * (out + g left + g^2 right)) and ProductVirtualRemainder (left * right * eq(tau, .), the eq table big-endian in tau, so the LAST
    tau is bound first). The final claim is the product form at the multilinear extensions' values at (r_0, ..., r_(v-1))."""
    P = pm.R_MOD
    rng = np.random.default_rng(23)
    to_int = lambda a: pm.from_mont(pm.from_limbs(a), P)
    to_mont = lambda v: ob.f_to_mont(FR, np.array([[(v >> (64 * i)) & (2**64 - 1) for i in range(4)]], dtype=np.uint64))[0]
    rnd = lambda n: ob.f_to_mont(FR, rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64))
    v, n = 5, 32

    def mle_partial(T, prefix, t, x):
        k = len(prefix)
        acc = 0
        for b in range(1 << (k + 1)):
            w = t if (b >> k) & 1 else (1 - t)
            for i in range(k):
                w = w * (prefix[i] if (b >> i) & 1 else (1 - prefix[i])) % P
            acc += w * T[b + (x << (k + 1))]
        return acc % P

    def round_from_definition(tables, F, prefix):
        k = len(prefix)
        return [sum(F([mle_partial(T, prefix, t, x) for T in tables]) for x in range(1 << (v - k - 1))) % P for t in range(4)]

    def prod(vals):
        out = 1
        for x in vals:
            out = out * x % P
        return out

    def drive(tables, F, make, round_evals, bind):
        ints = [[to_int(x) for x in t] for t in tables]
        claim = sum(F([T[j] for T in ints]) for j in range(n)) % P
        p = make(to_mont(claim))
        prefix = []
        for k in range(v):
            ev = [to_int(x) for x in round_evals(p)]
            assert ev == round_from_definition(ints, F, prefix), k
            ch = rnd(1)[0]
            bind(p, ch, ev)
            prefix.append(to_int(ch))
        return p, ints, prefix

    for three in (True, False):
        tabs = [rnd(n) for _ in range(3 if three else 2)]
        drive(tabs, prod, lambda c: ob.ValEvaluationProver(tabs[0], tabs[1], tabs[2] if three else None, c),
              lambda p: p.computeRoundPolynomial(), lambda p, ch, ev: p.bindChallengeWithPoly(ch, np.stack([to_mont(e) for e in ev])))
    tabs = [rnd(n) for _ in range(5)]
    drive(tabs, lambda a: a[0] * a[1] * (a[2] - a[3]) % P, lambda c: ob.OutputSumcheckProver(*tabs, c), lambda p: p.roundEvals(),
          lambda p, ch, ev: (p.bindChallenge(ch), p.updateClaim(np.stack([to_mont(e) for e in ev]), ch)))
    tabs = [rnd(n) for _ in range(4)]
    gamma = rnd(1)[0]
    g = to_int(gamma)

    def il_bind(p, ch, ev):
        p.bindChallenge(ch)
        p.updateClaim(np.stack([to_mont(e) for e in ev]), ch)
    drive(tabs, lambda a: a[0] * (a[1] + g * a[2] + g * g * a[3]) % P, lambda c: ob.InstructionLookupsClaimReduction(*tabs, gamma, c),
          lambda p: p.computeRoundPolynomialCubic(), il_bind)
    left, right, tau, kernel = rnd(n), rnd(n), rnd(v), rnd(1)[0]
    eq = ob.fr_eq_table(tau, kernel)  # index MSB <-> tau[0]: binding index bit 0 first binds tau[v-1] first

    def pr_bind(p, ch, ev):
        p.bindChallenge(ch)
        p.updateClaim(np.stack([to_mont(e) for e in ev]), ch)
    p, ints, prefix = drive([left, right, eq], prod, lambda c: ob.ProductRemainderProver(left, right, tau, kernel, c), lambda p: p.roundEvals(), pr_bind)
    # and the end of the protocol: the claim is the product of the three multilinear extensions at (r_0, ..., r_(v-1))
    full = [sum(prod([(prefix[i] if (idx >> i) & 1 else (1 - prefix[i])) for i in range(v)]) * T[idx] for idx in range(n)) % P for T in ints]
    assert prod(full) == to_int(p.current_claim)
    # the Lasso cycle phase folds the OTHER way (bindFirst order, prover.zig:411-441): variable k is index bit (v - 1 - k)
    lv = 4
    idx = np.zeros((1 << lv, 2), dtype=np.uint64)
    idx[:, 0] = rng.integers(0, 4, size=1 << lv, dtype=np.uint64)
    rr = rnd(lv)
    lp = ob.LassoProver(idx, lv, 2, rr)
    for k in range(2):
        lp.computeRoundPolynomial()
        lp.receiveChallenge(rnd(1)[0])
    table = [to_int(x) for x in lp.eq_evals[:1 << lv]]
    pre = []
    for k in range(lv):
        co = [to_int(x) for x in lp.computeRoundPolynomial()]
        half = 1 << (lv - k - 1)

        def at(t, x):  # eq_evals~ at (pre_0, ..., pre_(k-1), t, bits of x): variable m is index bit (lv - 1 - m), i.e. bit (k - m) of b
            acc = 0
            for b in range(1 << (k + 1)):
                w = t if b & 1 else 1 - t
                for m in range(k):
                    w = w * (pre[m] if (b >> (k - m)) & 1 else (1 - pre[m])) % P
                acc += w * table[(b << (lv - k - 1)) | x]
            return acc % P
        s0 = sum(at(0, x) for x in range(half)) % P
        s1 = sum(at(1, x) for x in range(half)) % P
        assert co[0] == s0 and (co[0] + co[1]) % P == s1, k
        ch = rnd(1)[0]
        lp.receiveChallenge(ch)
        pre.append(to_int(ch))
