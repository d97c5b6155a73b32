"""MSM parity: zg_msm_g1* (HIP) vs the CPU oracle (restated pippengerMSM), the golden
fixtures and the closed form. Everything goes through the C ABI."""
import os

import numpy as np
import pytest

from tests import util as U

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def zl():
    from zolt_amd import lib
    lib.init()
    return lib


@pytest.fixture(scope="module")
def ob():
    from oracle import binding
    return binding


@pytest.fixture(scope="module")
def gm(ob):
    return ob.g1_gen_multiples((1 << 16) + 1)  # P_i = (i+1)G, src/bench.zig:261-268


def _scalars(ob, seed, n):
    return ob.f_to_mont(ob.FR, U.random_raw256(seed, n))


def _check(zl, ob, xy, inf, sc, **cfg):
    b = zl.Bases.upload(xy, inf, **cfg)
    got, ginf = b.msm(sc)
    b.free()
    want, winf = ob.msm_g1(xy, inf, sc)
    assert ginf == winf
    assert np.array_equal(got, want), (got, want)


def test_golden_vectors(zl, ob):
    """tests/golden/vectors.json: duplicates, P/-P pairs, infinity bases, scalars 0/1/r-1."""
    from oracle import pymodel as pm
    for case in U.load_vectors()["msm"]:
        pts = [(int(p[0], 16), int(p[1], 16)) for p in case["points"]]
        xy, inf, sc = U.points_xy(pts), np.array(case["inf"], dtype=np.uint8), U.fr_hex(case["scalars"])
        for cfg in ({}, {"window_bits": 8, "precompute_levels": 1}, {"window_bits": 5, "precompute_levels": 3}):
            b = zl.Bases.upload(xy, inf, **cfg)
            got, ginf = b.msm(sc)
            b.free()
            want = None if case["result"] is None else tuple(int(h, 16) for h in case["result"])
            assert U.point_from_xy(got, ginf) == want, (case["n"], cfg)
            if want is None:
                assert not got.any()  # identity is {0,0,inf}, src/msm/mod.zig:24-30


@pytest.mark.parametrize("n", [0, 1, 7, 8, 9, 31, 32, 100, 1000, 2047, 2048, (1 << 13) - 1, 1 << 13, (1 << 15) - 1, 1 << 15, (1 << 16) + 1])
def test_sizes_vs_oracle(zl, ob, gm, n):
    """SURVEY §8(d) adversarial size set (the reference switches algorithm at n=8 and window size at 32/128/.../32768), plus the
    sizes at which the automatic plan here changes its window (64, 2048, 8192, 32768)."""
    _check(zl, ob, gm[:n], None, _scalars(ob, 1000 + n, n))


@pytest.mark.parametrize("cfg", [dict(window_bits=8, precompute_levels=1), dict(window_bits=13, precompute_levels=1),
                                 dict(window_bits=16, precompute_levels=1), dict(window_bits=11, precompute_levels=4),
                                 dict(window_bits=16, precompute_levels=16), dict(window_bits=7, precompute_levels=0)])
def test_window_and_precompute_configs(zl, ob, gm, cfg):
    """The result must not depend on the window size or the precompute depth."""
    n = 5000
    _check(zl, ob, gm[:n], None, _scalars(ob, 77, n), **cfg)


def test_adversarial_inputs(zl, ob, gm):
    from oracle import pymodel as pm
    n = 4096
    rng = np.random.default_rng(5)
    # all-equal points: every bucket add after the first is P+P or kP+P (forces the doubling branch)
    same = np.repeat(gm[3:4], n, axis=0)
    _check(zl, ob, same, None, _scalars(ob, 5, n))
    # all scalars equal: single-bucket pile-up per window
    one_scalar = np.repeat(_scalars(ob, 6, 1), n, axis=0)
    _check(zl, ob, gm[:n], None, one_scalar)
    _check(zl, ob, same, None, one_scalar)
    # P / -P pairs with equal scalars cancel to the identity
    neg = gm[:n].copy()
    neg[:, 4:] = ob.f_neg(ob.FP, gm[:n, 4:])
    xy = np.concatenate([gm[:n], neg])
    sc = _scalars(ob, 7, n)
    b = zl.Bases.upload(xy)
    got, ginf = b.msm(np.concatenate([sc, sc]))
    b.free()
    assert ginf == 1 and not got.any()
    # scalars 0, 1, r-1 and duplicates of the same (P, s)
    special = U.fr([0, 1, pm.R_MOD - 1, 2, pm.R_MOD - 2])
    sc = special[rng.integers(0, 5, size=n)]
    _check(zl, ob, gm[:n], None, sc)
    dup_idx = rng.integers(0, 16, size=n)
    _check(zl, ob, gm[dup_idx], None, _scalars(ob, 8, 16)[dup_idx])
    # infinity bases are skipped (src/msm/mod.zig:407), zero scalars -> identity (:875-891)
    inf = (rng.integers(0, 3, size=n) == 0).astype(np.uint8)
    _check(zl, ob, gm[:n], inf, _scalars(ob, 9, n))
    b = zl.Bases.upload(gm[:10])
    got, ginf = b.msm(np.zeros((10, 4), dtype=np.uint64))
    assert ginf == 1 and not got.any()
    got, ginf = b.msm(np.zeros((0, 4), dtype=np.uint64), n=0)  # empty -> identity (:938-947)
    assert ginf == 1 and not got.any()
    b.free()


def test_bench_family_closed_form(zl, ob, gm):
    """Bases (i+1)G, scalars 7i+13 (src/bench.zig:261-268): GPU == oracle == (sum s_i (i+1))·G."""
    from oracle import pymodel as pm
    for n in (16, 64, 256, 1 << 16):
        sc = ob.f_from_u64(ob.FR, np.arange(n, dtype=np.uint64) * np.uint64(7) + np.uint64(13))
        b = zl.Bases.upload(gm[:n])
        got, ginf = b.msm(sc)
        b.free()
        want = pm.msm_generator_multiples(range(1, n + 1), [7 * i + 13 for i in range(n)])
        assert U.point_from_xy(got, ginf) == want
    for case in U.load_vectors()["msm_bench_family"]:
        n = case["n"]
        sc = ob.f_from_u64(ob.FR, np.arange(n, dtype=np.uint64) * np.uint64(7) + np.uint64(13))
        b = zl.Bases.upload(gm[:n])
        got, ginf = b.msm(sc)
        b.free()
        assert U.point_from_xy(got, ginf) == tuple(int(h, 16) for h in case["result"])


def test_subrange_batch_and_partials(zl, ob, gm):
    n = 3000
    b = zl.Bases.upload(gm[:n])
    sc = _scalars(ob, 31, n)
    # MSM over bases[off..off+m) — HyperKZG.open commits to halving prefixes (commitment/mod.zig:287-315)
    for off, m in ((0, 1500), (100, 900), (2999, 1), (0, 0)):
        got, ginf = b.msm(sc[:m], off=off, n=m)
        want, winf = ob.msm_g1(gm[off:off + m], None, sc[:m])
        assert ginf == winf and np.array_equal(got, want)
    # BatchMSM.compute / batchCommit[i] == commit(poly_i) (commitment/mod.zig:1392-1420)
    batches = [_scalars(ob, 40 + k, n) for k in range(3)]
    outs, infs = b.msm_batch(batches)
    wouts, winfs = ob.msm_g1_batch(gm[:n], None, batches)
    assert np.array_equal(outs, wouts) and np.array_equal(infs, winfs)
    b.free()


@pytest.mark.parametrize("n,k,cfg", [(8, 5, {}), (200, 7, {}), (1024, 33, {}), (3000, 12, {}), (4096, 9, {"window_bits": 6}),
                                      (700, 4, {"window_bits": 8, "precompute_levels": 5}), (64, 300, {}),
                                      (5000, 3, {"window_bits": 16})])
def test_fused_batch_equals_separate_msms(zl, ob, gm, n, k, cfg):
    """zg_msm_g1_batch with short vectors runs ONE launch set with k times the bucket groups (BatchMSM.compute,
    src/msm/mod.zig:545-565; Dory rows, src/poly/commitment/dory.zig:646-670): every result must be the oracle's MSM
    of that vector — including all-zero vectors, vectors of ones, repeated vectors, infinity bases, a batch larger than
    one launch set takes (split), and the unfused rotation used for wide windows."""
    inf = np.zeros(n, dtype=np.uint8)
    inf[::7] = 1
    b = zl.Bases.upload(gm[:n], inf, **cfg)
    batches = [_scalars(ob, 7000 + 31 * n + j, n) for j in range(k)]
    batches[0] = np.zeros((n, 4), dtype=np.uint64)                       # identity result
    if k > 2:
        batches[2] = ob.f_from_u64(ob.FR, np.ones(n, dtype=np.uint64))   # one bucket per window takes every point
    if k > 3:
        batches[3] = batches[1].copy()                                  # equal vectors give equal commitments
    outs, infs = b.msm_batch(batches)
    for j in range(k):
        want, winf = ob.msm_g1(gm[:n], inf, batches[j])
        assert infs[j] == winf and np.array_equal(outs[j], want), (n, k, j)
    assert infs[0] == 1
    # a second call reuses the cached fused workspace; a call with fewer vectors fits the same workspace
    outs2, infs2 = b.msm_batch(batches[:max(2, k // 2)])
    assert np.array_equal(outs2, outs[:max(2, k // 2)]) and np.array_equal(infs2, infs[:max(2, k // 2)])
    # a plain MSM on the same handle afterwards is unaffected
    got, ginf = b.msm(batches[1])
    assert ginf == infs[1] and np.array_equal(got, outs[1])
    b.free()


def test_fused_batch_dev_and_prefix(zl, ob, gm):
    """Device-resident form: vectors back to back in HBM, records (xy[8], flag word) written to HBM; vectors shorter than
    the handle commit to the prefix bases[0..n) (HyperKZG.commit: n = min(evals.len, srs.len), commitment/mod.zig:246)."""
    import ctypes as C
    N, n, k = 2048, 500, 6
    b = zl.Bases.upload(gm[:N])
    sc = np.concatenate([_scalars(ob, 7700 + j, n) for j in range(k)])
    d_sc, d_out = C.c_void_p(), C.c_void_p()
    assert zl._lib.zg_dev_alloc(C.c_size_t(sc.size * 8), C.byref(d_sc)) == 0
    assert zl._lib.zg_dev_alloc(C.c_size_t(k * 72), C.byref(d_out)) == 0
    assert zl._lib.zg_memcpy_h2d(d_sc, sc.ctypes.data_as(C.c_void_p), C.c_size_t(sc.size * 8)) == 0
    b.msm_batch_dev(d_sc.value, n, k, d_out.value)
    zl.sync()
    rec = np.empty((k, 9), dtype=np.uint64)
    assert zl._lib.zg_memcpy_d2h(rec.ctypes.data_as(C.c_void_p), d_out, C.c_size_t(k * 72)) == 0
    for j in range(k):
        want, winf = ob.msm_g1(gm[:n], None, sc[j * n:(j + 1) * n])
        assert int(rec[j, 8]) & 0xFF == winf and np.array_equal(rec[j, :8], want)
    zl._lib.zg_dev_free(d_sc)
    zl._lib.zg_dev_free(d_out)
    b.free()


def test_parallel_msm_partials_combine(zl, ob, gm):
    """ParallelMSM (src/msm/mod.zig:572-680): contiguous chunks -> Jacobian partials -> serial combine.
    Partials must equal fromAffine(SingleMSM.compute(chunk)) exactly, the combined point the full MSM."""
    import ctypes as C
    n, T = 8192, 4
    sc = _scalars(ob, 55, n)
    chunk = (n + T - 1) // T
    d_sc = C.c_void_p()
    d_part = C.c_void_p()
    assert zl._lib.zg_dev_alloc(C.c_size_t(n * 32), C.byref(d_sc)) == 0
    assert zl._lib.zg_dev_alloc(C.c_size_t(T * 96), C.byref(d_part)) == 0
    assert zl._lib.zg_memcpy_h2d(d_sc, sc.ctypes.data_as(C.c_void_p), C.c_size_t(n * 32)) == 0
    b = zl.Bases.upload(gm[:n])
    for t in range(T):
        b.msm_partial_dev(d_sc.value + t * chunk * 32, chunk, d_part.value + t * 96, off=t * chunk)
    zl.sync()
    parts = np.empty((T, 12), dtype=np.uint64)
    assert zl._lib.zg_memcpy_d2h(parts.ctypes.data_as(C.c_void_p), d_part, C.c_size_t(T * 96)) == 0
    one = ob.f_from_u64(ob.FP, np.array([1], dtype=np.uint64))[0]
    for t in range(T):
        want, winf = ob.msm_g1(gm[t * chunk:(t + 1) * chunk], None, sc[t * chunk:(t + 1) * chunk])
        assert winf == 0 and np.array_equal(parts[t, :8], want) and np.array_equal(parts[t, 8:], one)
    got, ginf = zl.combine_partials_dev(d_part.value, T)
    want, winf = ob.msm_g1_parallel(gm[:n], None, sc, T)
    assert ginf == winf and np.array_equal(got, want)
    full, finf = b.msm(sc)
    assert np.array_equal(full, got)
    # the un-normalised partials (no per-GPU inversion) are other Jacobian representatives of the same points,
    # and combine to the same bytes
    for t in range(T):
        b.msm_partial_fast_dev(d_sc.value + t * chunk * 32, chunk, d_part.value + t * 96, off=t * chunk)
    zl.sync()
    fast = np.empty((T, 12), dtype=np.uint64)
    assert zl._lib.zg_memcpy_d2h(fast.ctypes.data_as(C.c_void_p), d_part, C.c_size_t(T * 96)) == 0
    for t in range(T):
        a_xy, a_inf = ob.g1_jac_to_affine(fast[t])
        assert a_inf == 0 and np.array_equal(a_xy, parts[t, :8])
    got2, ginf2 = zl.combine_partials_dev(d_part.value, T)
    assert ginf2 == ginf and np.array_equal(got2, got)
    # more partials than lanes in the combine wave, with identities mixed in
    many = np.concatenate([np.tile(fast, (20, 1)), np.tile(parts, (3, 1))])  # 92 records
    ident = np.concatenate([one, one, np.zeros(4, dtype=np.uint64)])
    many[5] = ident
    many[77] = ident
    d_many = C.c_void_p()
    assert zl._lib.zg_dev_alloc(C.c_size_t(many.size * 8), C.byref(d_many)) == 0
    assert zl._lib.zg_memcpy_h2d(d_many, many.ctypes.data_as(C.c_void_p), C.c_size_t(many.size * 8)) == 0
    gm_xy, gm_inf = zl.combine_partials_dev(d_many.value, many.shape[0])
    acc = ident.copy()
    for rec in many:
        acc = ob.g1_jac_add(acc, rec)
    w_xy, w_inf = ob.g1_jac_to_affine(acc)
    assert gm_inf == w_inf and np.array_equal(gm_xy, w_xy)
    zl._lib.zg_dev_free(d_many)
    # an empty shard contributes the reference's identity record (1,1,0)
    b.msm_partial_dev(d_sc.value, 0, d_part.value)
    zl.sync()
    assert zl._lib.zg_memcpy_d2h(parts.ctypes.data_as(C.c_void_p), d_part, C.c_size_t(96)) == 0
    assert np.array_equal(parts[0, :4], one) and np.array_equal(parts[0, 4:8], one) and not parts[0, 8:].any()
    b.free()
    zl._lib.zg_dev_free(d_sc)
    zl._lib.zg_dev_free(d_part)


def test_srs_and_reference_proof_commitment(zl, ob, golden_dir):
    """HyperKZG.setup (mock SRS) + commit of fibonacci.elf's bytecode polynomial reproduce the
    bytes stored in the reference's captured proof (logs/zolt_proof_regular.bin @8..72)."""
    from oracle import pymodel as pm
    elf = open(os.path.join(golden_dir, "fibonacci.elf"), "rb").read()
    proof = open(os.path.join(golden_dir, "zolt_proof_regular.bin"), "rb").read()
    n = 128
    g = U.points_xy([pm.G1])
    taus = U.fr([pow(pm.TAU, i, pm.R_MOD) for i in range(n)])
    srs, sinf = zl.g1_scalar_mul_batch(np.repeat(g, n, axis=0), np.zeros(n, dtype=np.uint8), taus)
    wsrs, winf = ob.hyperkzg_setup(n)
    assert np.array_equal(srs, wsrs) and np.array_equal(sinf, winf)
    ev = np.zeros(n, dtype=np.uint64)
    ev[:104] = np.frombuffer(elf[0x1000:0x1000 + 104], dtype=np.uint8)
    b = zl.Bases.upload(srs, sinf)
    got, ginf = b.msm(U.fr(ev))
    b.free()
    assert ginf == 0
    from zolt_amd import api
    assert api.commitment_to_bytes(got, ginf) == api.parse_zolt_proof_commitments(proof)["bytecode.commitment"] == proof[8:72]
    # scalarMul edge cases: 0*P = inf, 1*P = P, k*inf = inf (src/msm/mod.zig:503-505,802-825)
    out, oinf = zl.g1_scalar_mul_batch(np.repeat(g, 3, axis=0), np.array([0, 0, 1], dtype=np.uint8), U.fr([0, 1, 5]))
    assert list(oinf) == [1, 0, 1] and np.array_equal(out[1], g[0]) and not out[0].any() and not out[2].any()


def test_all_commitments_of_the_reference_proof_header(zl, ob, golden_dir):
    """BASELINE config 5 as far as it goes without a Zig toolchain: the three commitments a `zolt prove examples/fibonacci.elf` run
    stores in the ZOLT v1 proof header (src/zkvm/serialization.zig:283-306) — bytecode @8, memory @232 (identity: empty RAM
    trace), registers @488 — produced by the GPU path (HyperKZG.setup with the fixed-base kernel at the reference's srs_size =
    1280, HyperKZG.commit on the resident SRS) and compared with the reference's own captured bytes."""
    from zolt_amd import api
    elf = open(os.path.join(golden_dir, "fibonacci.elf"), "rb").read()
    proof = open(os.path.join(golden_dir, "zolt_proof_regular.bin"), "rb").read()
    hdr = api.parse_zolt_proof_commitments(proof)
    params = api.HyperKZG.setup(1280)  # logs/zolt.log:10
    try:
        bc = np.zeros(128, dtype=np.uint64)
        bc[:104] = np.frombuffer(elf[0x1000:0x1000 + 104], dtype=np.uint8)  # commitBytecode, src/zkvm/mod.zig:1519-1547
        reg = np.zeros(256, dtype=np.uint64)
        reg[:54] = np.array(U.fibonacci_rd_values(elf), dtype=np.uint64)    # commitRegisters, :1585-1617
        polys = [U.fr(bc), np.zeros((0, 4), dtype=np.uint64), U.fr(reg)]
        got = api.HyperKZG.batchCommit(params, polys)
        for (xy, fl), name in zip(got, ("bytecode.commitment", "memory.commitment", "register.commitment")):
            assert api.commitment_to_bytes(xy, fl) == hdr[name], name
        assert hdr["register.commitment"] == proof[488:552] and hdr["register.commitment"] != bytes(64)
        one = api.HyperKZG.commit(params, polys[2])
        assert api.commitment_to_bytes(*one) == proof[488:552]
        # the same commitments from the MACHINE WORDS (zg_msm_g1_u64: 8 bytes per evaluation across PCIe, fromU64 on the device)
        assert api.commitment_to_bytes(*api.HyperKZG.commitU64(params, bc)) == hdr["bytecode.commitment"]
        assert api.commitment_to_bytes(*api.HyperKZG.commitU64(params, reg)) == proof[488:552]
        assert api.commitment_to_bytes(*api.HyperKZG.commitU64(params, np.zeros(0, dtype=np.uint64))) == bytes(64)
        # the proof header a `zolt prove` run would write from these commitments: byte-identical to the captured file's first 744 bytes
        header = api.serialize_zolt_proof_header({"bytecode.commitment": got[0], "memory.commitment": got[1], "register.commitment": got[2]})
        assert len(header) == 744 and header == proof[:744]
    finally:
        params.deinit()


@pytest.mark.parametrize("n", [1 << 20, (1 << 20) + 1, 1 << 22])
def test_full_size_closed_form(zl, ob, n):
    """BASELINE config 2 (2^20 points, one GPU), the 2^20 + 1 member of SURVEY §8(d)'s adversarial size set and the metric's
    second size 2^22: uniform scalars; the result must equal the size-independent closed form (sum s_i (i+1) mod r)·G.
    At 2^22 the same inputs are also run as BASELINE config 4's shards (ParallelMSM's contiguous chunks for G = 2 and 8 ranks,
    one handle per shard, un-normalised Jacobian partials, device combine) — identical bytes for every G."""
    from oracle import pymodel as pm
    gm = ob.g1_gen_multiples(n)
    raw = U.random_raw256(0x5A4F4C54, n)
    sc = ob.f_to_mont(ob.FR, raw)
    b = zl.Bases.upload(gm)
    # the automatic plan (the GPU's optimalWindowSize): 17-bit windows while the 15 n table rows fit the 26-bit reference that
    # leaves the two-pass sort 5 fine key bits (2^22 points included), else 16
    assert b.plan() == ((17, 15, 15) if n * 15 <= 1 << 26 else (16, 16, 16))
    got, ginf = b.msm(sc)
    got2, ginf2 = b.msm(sc)  # idempotent: same handle, same answer
    b.free()
    assert np.array_equal(got, got2) and ginf == ginf2 == 0
    # closed form in Python big ints, vectorised per limb to keep it O(seconds)
    k = np.arange(1, n + 1, dtype=object)
    tot = 0
    for limb in range(4):
        tot += int((raw[:, limb].astype(object) * k).sum()) << (64 * limb)
    # raw values may exceed r: the scalar is raw mod r
    want = pm.ec_mul(tot % pm.R_MOD, pm.G1)
    assert U.point_from_xy(got, ginf) == want
    if n == 1 << 22:
        import ctypes as C
        from zolt_amd import api
        d_sc, d_part = C.c_void_p(), C.c_void_p()
        assert zl._lib.zg_dev_alloc(C.c_size_t(n * 32), C.byref(d_sc)) == 0
        assert zl._lib.zg_dev_alloc(C.c_size_t(8 * 96), C.byref(d_part)) == 0
        assert zl._lib.zg_memcpy_h2d(d_sc, sc.ctypes.data_as(C.c_void_p), C.c_size_t(n * 32)) == 0
        for G in (2, 8):
            for rank, (s0, s1) in enumerate(api.shard_bounds(n, G)):
                shard = zl.Bases.upload(gm[s0:s1])
                shard.msm_partial_fast_dev(d_sc.value + s0 * 32, s1 - s0, d_part.value + rank * 96)
                zl.sync()
                shard.free()
            cxy, cinf = zl.combine_partials_dev(d_part.value, G)
            assert cinf == ginf and np.array_equal(cxy, got), G
        zl._lib.zg_dev_free(d_sc)
        zl._lib.zg_dev_free(d_part)
        # the same config through the ONE-process entry points (zg_g1_bases_upload_sharded / zg_msm_g1_sharded): eight logical shards
        # resident at once on this device, worker threads, the partial exchange and the device combine
        os.environ["ZG_SHARDS"] = "8"
        try:
            zl.init_devices(1)
            sb = zl.ShardedBases.upload(gm)
            assert len(sb.shards()) == 8
            sxy, sinf = sb.msm(sc)
            sb.free()
        finally:
            del os.environ["ZG_SHARDS"]
        assert sinf == ginf and np.array_equal(sxy, got)


@pytest.mark.parametrize("kind", ["boolean", "bytes", "all_equal", "one_hot_bucket", "mixed"])
def test_skewed_scalar_distributions(zl, ob, gm, kind):
    """Real witness columns are not uniform: 0/1 flags, bytes, constants. The chunk-scheduled accumulate must
    give the oracle's bytes whatever the bucket-size skew (heavy buckets go through the block-tree stages)."""
    n = 1 << 16
    rng = np.random.default_rng(77)
    if kind == "boolean":
        vals = rng.integers(0, 2, size=n)
    elif kind == "bytes":
        vals = rng.integers(0, 256, size=n)
    elif kind == "all_equal":
        vals = np.full(n, 0x1234)
    elif kind == "one_hot_bucket":
        vals = np.where(rng.integers(0, 10, size=n) == 0, rng.integers(0, 1 << 30, size=n), 7)
    else:
        vals = rng.integers(0, 1 << 62, size=n)
    sc = ob.f_from_u64(ob.FR, vals.astype(np.uint64))
    if kind == "mixed":  # a few full-width scalars and negated small ones on top
        sc[::17] = _scalars(ob, 99, len(sc[::17]))
        sc[5::29] = ob.f_neg(ob.FR, sc[5::29])
    _check(zl, ob, gm[:n], None, sc)
    _check(zl, ob, gm[:n], None, sc, window_bits=13, precompute_levels=1)


@pytest.mark.parametrize("c", list(range(2, 20)))
def test_every_window_size(zl, ob, gm, c):
    """every digit-kernel instantiation (window_bits 2..19), full precompute and none"""
    n = 2000
    sc = _scalars(ob, 4000 + c, n)
    _check(zl, ob, gm[:n], None, sc, window_bits=c, precompute_levels=0)
    W = (255 + c - 1) // c
    if W <= 64 and (1 << (c - 1)) * W <= 1 << 21:  # table-less: one bucket set per window, at most 2^21 buckets in all (c <= 18)
        _check(zl, ob, gm[:n], None, sc, window_bits=c, precompute_levels=1)


@pytest.mark.parametrize("logn", [17, 18, 19])
def test_auto_plan_sizes(zl, ob, logn):
    """the per-rank shard sizes of a sharded 2^20..2^22 MSM under the automatic plan"""
    n = 1 << logn
    gmn = ob.g1_gen_multiples(n)
    sc = _scalars(ob, 5000 + logn, n)
    b = zl.Bases.upload(gmn)
    assert b.plan() == (16, 16, 16)
    got, ginf = b.msm(sc)
    b.free()
    want, winf = ob.msm_g1_parallel(gmn, None, sc, 8)
    assert ginf == winf and np.array_equal(got, want)


@pytest.mark.parametrize("env", [{}, {"ZG_MSM_TWO_PASS_SORT": "0"}, {"ZG_MSM_FINE_BITS": "5", "ZG_MSM_FINE_BITS_MIN": "2"},
                                 {"ZG_MSM_FINE_BITS": "3", "ZG_MSM_FINE_BITS_MIN": "2", "ZG_MSM_TWO_PASS_SPAN": "256"},
                                 {"ZG_MSM_TWO_PASS_SPAN": "8192"}])
@pytest.mark.parametrize("wb", [16, 17])
def test_two_pass_sort_variants(zl, ob, gm, env, wb, monkeypatch):
    """The two-pass counting sort used for 2^15 buckets (coarse partition, then per-bin slices) against the single-pass sort and
    with other bin / block shapes — uniform scalars, a 0/1 column (half of all entries in ONE coarse bin, split over many
    slices), a constant column and infinity bases. Same bytes as the oracle every time."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    n = 40000
    rng = np.random.default_rng(5)
    inf = np.zeros(n, dtype=np.uint8)
    inf[3::13] = 1
    b = zl.Bases.upload(gm[:n], inf, window_bits=wb)  # 2^15 / 2^16 buckets
    cases = [_scalars(ob, 31337, n), ob.f_from_u64(ob.FR, rng.integers(0, 2, size=n).astype(np.uint64)),
             ob.f_from_u64(ob.FR, np.full(n, 0xABCDEF, dtype=np.uint64))]
    for sc in cases:
        got, ginf = b.msm(sc)
        want, winf = ob.msm_g1(gm[:n], inf, sc)
        assert ginf == winf and np.array_equal(got, want)
    got, ginf = b.msm(cases[0][:17000], off=16500, n=17000)  # a sub-range that does not fit the side table
    want, winf = ob.msm_g1(gm[16500:33500], inf[16500:33500], cases[0][:17000])
    assert ginf == winf and np.array_equal(got, want)
    b.free()


@pytest.mark.parametrize("env", [{}, {"ZG_MSM_SIDE_TABLE": "0"}, {"ZG_MSM_BATCH_FUSE": "0"}])
def test_side_table_routing(zl, ob, gm, env, monkeypatch):
    """A wide-window handle (n >= 32768) answers MSMs over a short prefix, and batches of them, from its narrow-window side
    table of the first 16384 bases; ranges that leave the prefix use the main table. Same bytes either way, and with the
    side table or the fusing switched off."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    n = 40000
    inf = np.zeros(n, dtype=np.uint8)
    inf[5::11] = 1
    b = zl.Bases.upload(gm[:n], inf)
    sc = _scalars(ob, 4321, n)
    for off, m in ((0, 16384), (0, 1000), (384, 16000), (1, 16384), (16000, 500), (0, 16385), (0, n)):
        got, ginf = b.msm(sc[:m], off=off, n=m)
        want, winf = ob.msm_g1(gm[off:off + m], inf[off:off + m], sc[:m])
        assert ginf == winf and np.array_equal(got, want), (off, m)
    batches = [_scalars(ob, 4400 + j, 3000) for j in range(5)]
    outs, infs = b.msm_batch(batches)
    for j in range(5):
        want, winf = ob.msm_g1(gm[:3000], inf[:3000], batches[j])
        assert infs[j] == winf and np.array_equal(outs[j], want)
    b.free()


@pytest.mark.parametrize("c", [11, 12, 13, 16, 17])
@pytest.mark.parametrize("env", [{}, {"ZG_MSM_REDUCE_2D": "0"}])
def test_bucket_reduction_rows_and_columns(zl, ob, gm, c, env, monkeypatch):
    """The two-dimensional bucket reduction (row / column sums of the 2^hb x 2^lb bucket matrix, then bit sums over rows and
    columns) against the oracle with digit magnitudes chosen to sit on the matrix edges: every magnitude 1..2^(c-1) for the
    narrow windows, and for c = 16 all multiples of 2^lb, their neighbours, the first row, the first column and the top
    bucket NB itself (the only one outside the matrix) — each in the lowest window, plus one uniform vector."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    NB, lb = 1 << (c - 1), c // 2
    if c <= 13:
        mags = np.arange(1, NB + 1, dtype=np.uint64)
    else:
        edge = np.arange(0, NB + 1, 1 << lb, dtype=np.int64)
        mags = np.unique(np.concatenate([edge, edge - 1, edge + 1, np.arange(1, (1 << lb) + 2), [NB - 1, NB]]))
        mags = mags[(mags >= 1) & (mags <= NB)].astype(np.uint64)
    n = len(mags)
    assert n <= len(gm)
    b = zl.Bases.upload(gm[:n], None, window_bits=c)
    # magnitude m as a positive digit (scalar m) and as a negative one (scalar 2^c - m: digit -m with a carry into window 1)
    for vals in (mags, (np.uint64(1 << c) - mags)):
        sc = ob.f_from_u64(ob.FR, vals)
        got, ginf = b.msm(sc)
        want, winf = ob.msm_g1(gm[:n], None, sc)
        assert ginf == winf and np.array_equal(got, want)
    sc = _scalars(ob, 9100 + c, n)
    got, ginf = b.msm(sc)
    want, winf = ob.msm_g1(gm[:n], None, sc)
    assert ginf == winf and np.array_equal(got, want)
    b.free()


@pytest.mark.parametrize("env", [{"ZG_MSM_CHUNK_SCHED": "0"}, {"ZG_MSM_LDS_SORT": "0"}, {"ZG_MSM_LANES": "1"}, {"ZG_MSM_COMBINE_PER_QUAD": "8"},
                                 {"ZG_MSM_CHUNK_THREADS": "1000"}, {"ZG_MSM_CHUNK_SCHED": "0", "ZG_MSM_LDS_SORT": "0", "ZG_MSM_SLICES": "4"}])
def test_alternate_code_paths(zl, ob, gm, env, monkeypatch):
    """the fallback schedulers (per-bucket lanes, global-atomic counting sort) and odd tuning values stay bit-exact"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    n = 6000
    rng = np.random.default_rng(11)
    sc = _scalars(ob, 777, n)
    sc[rng.integers(0, n, size=500)] = U.fr([3])[0]  # some skew
    _check(zl, ob, gm[:n], None, sc)
    _check(zl, ob, gm[:n], None, sc, window_bits=16, precompute_levels=2)


@pytest.mark.parametrize("env", [{"ZG_MSM_PRECOMPUTE_V1": "1"}, {"ZG_MSM_ROWCOL_WAVE_FROM": "1"}, {"ZG_MSM_ROWCOL_WAVE_FROM": "0"},
                                 {"ZG_MSM_ROWS_SHARED_TAIL": "0"}, {"ZG_MSM_SIDE_TABLE": "0"}])
def test_round5_alternate_code_paths(zl, ob, gm, env, monkeypatch):
    """the round-4 table kernel (an inversion per level, no records), row / column sums by one wave per row for a single bucket set and
    never, a reduction per row of a table-less batch, no side table: every one of them the same bytes — on a handle with its table
    (20 000 bases: the side table rides in the table kernel's launch), on one planned for a few uses, for single MSMs and batches."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    n = 20000
    sc = _scalars(ob, 779, n)
    want, winf = ob.msm_g1(gm[:n], None, sc)
    for uses in (0, 2):
        b = zl.Bases.upload(gm[:n], None, expected_uses=uses)
        got, ginf = b.msm(sc)
        assert ginf == winf and np.array_equal(got, want), (env, uses)
        short = 3000  # a prefix the side table serves (when there is one)
        g2, i2 = b.msm(sc[:short], n=short)
        w2, wi2 = ob.msm_g1(gm[:short], None, sc[:short])
        assert i2 == wi2 and np.array_equal(g2, w2), (env, uses, "short")
        rows = [_scalars(ob, 780 + j, n) for j in range(3)]
        outs, infs = b.msm_batch(rows, n=n)
        for j in range(3):
            wj, wij = ob.msm_g1(gm[:n], None, rows[j])
            assert infs[j] == wij and np.array_equal(outs[j], wj), (env, uses, "batch", j)
        b.free()


def test_affine_point_add_and_double(zl, ob, gm):
    """AffinePoint.add / double (src/msm/mod.zig:74-138) as zg_g1_affine_add_batch: generic pairs, P + P (the doubling branch),
    P + (-P) (identity), identity operands, and the reference's KAT 'generator affine-double == Jacobian-double'
    (src/poly/commitment/mod.zig:1240-1258), all against the oracle's restatement of the same functions."""
    from zolt_amd import api
    n = 512
    a = gm[:n].copy()
    b = gm[n:2 * n].copy()
    b[:8] = a[:8]  # P + P
    neg = a[8:16].copy()
    pm_p = 21888242871839275222246405745257275088696311157297823662689037894645226208583
    for i in range(8):  # P + (-P): y -> p - y (Montgomery form is linear)
        y = sum(int(v) << (64 * k) for k, v in enumerate(neg[i, 4:]))
        neg[i, 4:] = [((pm_p - y) >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)]
    b[8:16] = neg
    ai = np.zeros(n, dtype=np.uint8)
    bi = np.zeros(n, dtype=np.uint8)
    ai[16:20] = 1
    bi[18:24] = 1
    out, oinf = zl.g1_affine_add_batch(a, ai, b, bi)
    for i in range(n):
        w, wi = ob.g1_add_affine(a[i], int(ai[i]), b[i], int(bi[i]))
        assert oinf[i] == wi and (wi or np.array_equal(out[i], w)), i
    assert oinf[8:16].all() and not out[8:16].any()  # identity is written as x = y = 0, inf = 1
    d, di = api.AffinePoint.double(api.generator())
    w, wi = ob.g1_double_affine(api.generator(), 0)
    assert di == wi == 0 and np.array_equal(d, w)
    jd = ob.g1_jac_to_affine(ob.g1_jac_double(np.concatenate([api.generator(), api.fp_from_int(1)])))
    assert np.array_equal(d, jd[0])
    s2, s2i = api.AffinePoint.add(api.generator(), 0, api.generator(), 0)
    assert s2i == 0 and np.array_equal(s2, d) and api.AffinePoint.isOnCurve(d)
    z, zi = api.AffinePoint.double(np.zeros(8, dtype=np.uint64), 1)
    assert zi == 1


def test_mock_srs_and_scalar_mul_at_reference_default_size(zl, ob):
    """HyperKZG.setup at the reference's default srs_size = 1280 (logs/zolt.log:10-12; src/poly/commitment/mod.zig:174-213):
    the device batch scalar multiplication equals the oracle's mock SRS point for point."""
    from zolt_amd import api
    n = 1280
    params = api.HyperKZG.setup(n)
    wsrs, winf = ob.hyperkzg_setup(n)
    assert np.array_equal(params.powers_of_tau_g1, wsrs) and np.array_equal(params.infinity, winf)
    assert zl.g1_is_on_curve_batch(params.powers_of_tau_g1, params.infinity).all()
    params.deinit()


@pytest.mark.parametrize("slices", [2, 3, 8])
def test_host_scalar_path_sliced(zl, ob, gm, slices, monkeypatch):
    """zg_msm_g1 with host scalars cuts long vectors into slices (copy of slice i under the launch set of slice i-1, partials
    combined on the device): same bytes as the unsliced call and the oracle, including a sub-range and infinity bases."""
    monkeypatch.setenv("ZG_MSM_HOST_SLICES", str(slices))
    monkeypatch.setenv("ZG_MSM_HOST_SLICE_MIN", "1000")
    n = 5003
    inf = np.zeros(n, dtype=np.uint8)
    inf[3::50] = 1
    sc = ob.f_to_mont(ob.FR, U.random_raw256(4000 + slices, n))
    b = zl.Bases.upload(gm[:n], inf)
    try:
        want = ob.msm_g1(gm[:n], inf, sc)
        got = b.msm(sc)
        assert got[1] == want[1] and np.array_equal(got[0], want[0])
        w2 = ob.msm_g1(gm[100:4100], inf[100:4100], sc[:4000])
        g2 = b.msm(sc[:4000], off=100, n=4000)
        assert g2[1] == w2[1] and np.array_equal(g2[0], w2[0])
        z = b.msm(np.zeros((n, 4), dtype=np.uint64))
        assert z[1] == 1
        # fewer scalars than slices at the END of the bases: the empty trailing slices still hand in their identity records
        # (found by tools/fuzz_msm.py in round 4: their start lay beyond the uploaded bases and the call was refused)
        monkeypatch.setenv("ZG_MSM_HOST_SLICE_MIN", "1")
        m = slices - 1
        w3 = ob.msm_g1(gm[n - m:n], inf[n - m:n], sc[:m])
        g3 = b.msm(sc[:m], off=n - m, n=m)
        assert g3[1] == w3[1] and np.array_equal(g3[0], w3[0])
    finally:
        b.free()


@pytest.mark.parametrize("kind", ["boolean", "bytes", "one_hot_bucket"])
def test_skewed_scalars_in_point_slices(zl, ob, gm, kind, monkeypatch):
    """skewed columns (a 0/1 flag puts half of all points into one bucket: the heavy / huge bucket stages) with the launch set cut into
    eight point slices: every slice runs the heavy stages on its own bucket set and reduction state, the sets are added up — the
    oracle's bytes"""
    monkeypatch.setenv("ZG_MSM_TABLE_SPAN_MB", "1")
    monkeypatch.setenv("ZG_MSM_TABLE_SPAN_MIN_POINTS", "8192")
    n = 1 << 16
    rng = np.random.default_rng(78)
    if kind == "boolean":
        vals = rng.integers(0, 2, size=n)
    elif kind == "bytes":
        vals = rng.integers(0, 256, size=n)
    else:
        vals = np.where(rng.integers(0, 10, size=n) == 0, rng.integers(0, 1 << 30, size=n), 7)
    sc = ob.f_from_u64(ob.FR, vals.astype(np.uint64))
    _check(zl, ob, gm[:n], None, sc)
    _check(zl, ob, gm[:n], None, sc, window_bits=13, precompute_levels=1)


def test_point_slices_with_slice_local_references(zl, ob, monkeypatch):
    """2^21 + 12345 points (a 2 GiB table) cut into five slices of ~420 k points: a slice's sorted references are level << 19 | point
    instead of table rows, so the slice sorts with 7 fine bits where the handle's own plan has 5 (slice_sort_plan), and the accumulate
    kernel decodes them. Checked by the closed form: bases (i+1)*G, so the MSM is (sum s_i (i+1))*G — one scalar multiplication on an
    independent kernel path; also with the slice plan switched off, a sub-range that is not sliced, and 0/1 scalars."""
    import torch
    from zolt_amd import api
    monkeypatch.setenv("ZG_MSM_TABLE_SPAN_MB", "512")
    n = (1 << 21) + 12345
    g = api.generator()
    ks = np.zeros((n, 4), dtype=np.uint64)
    ks[:, 0] = np.arange(1, n + 1, dtype=np.uint64)
    bases, _ = zl.g1_scalar_mul_batch(np.repeat(g[None, :], n, axis=0), np.zeros(n, dtype=np.uint8), zl.field_op(zl.FR, zl.OP_TO_MONT, ks))
    rng = np.random.default_rng(2121)
    small = rng.integers(0, 1 << 40, size=n, dtype=np.uint64).astype(object)
    big = [int(x) for x in rng.integers(0, 1 << 62, size=n, dtype=np.uint64)]
    vals = [(int(a) << 180) + c for a, c in zip(small, big)]  # 220-bit scalars: every window is used
    raw = np.array([[(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)] for v in vals], dtype=np.uint64)
    sc = zl.field_op(zl.FR, zl.OP_TO_MONT, raw)
    want_k = sum(v * (i + 1) for i, v in enumerate(vals)) % api.R_MOD
    want = api.MSM.scalarMul(g, api.fr_from_int(want_k))
    b = zl.Bases.upload(bases)
    try:
        assert b.plan()[0] >= 16
        got = b.msm(sc)
        assert got[1] == want[1] and np.array_equal(got[0], want[0])
        monkeypatch.setenv("ZG_MSM_SLICE_LOCAL_REFS", "0")
        got = b.msm(sc)
        assert got[1] == want[1] and np.array_equal(got[0], want[0])
        monkeypatch.delenv("ZG_MSM_SLICE_LOCAL_REFS")
        m = 700001  # a sub-range below two slices: one launch set with table-row references on the same workspaces
        sub_k = sum(v * (i + 1 + 1000) for i, v in enumerate(vals[:m])) % api.R_MOD
        wsub = api.MSM.scalarMul(g, api.fr_from_int(sub_k))
        gsub = b.msm(sc[:m], off=1000, n=m)
        assert gsub[1] == wsub[1] and np.array_equal(gsub[0], wsub[0])
        flags = rng.integers(0, 2, size=n)
        fsc = ob.f_from_u64(ob.FR, flags.astype(np.uint64))
        wf = api.MSM.scalarMul(g, api.fr_from_int(int(sum(int(i + 1) for i in np.nonzero(flags)[0])) % api.R_MOD))
        gf = b.msm(fsc)
        assert gf[1] == wf[1] and np.array_equal(gf[0], wf[0])
    finally:
        b.free()


def test_sub_range_on_a_handle_whose_slices_alone_sort_in_two_passes(zl, ob):
    """2^23 points with 17-bit windows: a table row index (15 * 2^23 rows) leaves the handle's own plan no room for the two-pass sort's
    fine key bits and 2^16 buckets do not fit the LDS sort, so unsliced launch sets sort through global atomics — while the point
    slices of a full-size MSM sort in two passes on slice-local references and own the per-block histogram buffer. A short sub-range
    (HyperKZG.open's prefixes, MSM.compute on a slice of the SRS) must take the global path, not mistake that buffer for the LDS sort's
    (round-3 advisor finding: a division by zero in the launch geometry). Closed form: bases (i+1) G, so MSM = (sum s_i (i+1)) G."""
    from zolt_amd import api
    n = 1 << 23
    g = api.generator()
    ks = np.zeros((n, 4), dtype=np.uint64)
    ks[:, 0] = np.arange(1, n + 1, dtype=np.uint64)
    bases, _ = zl.g1_scalar_mul_batch(np.repeat(g[None, :], n, axis=0), np.zeros(n, dtype=np.uint8), zl.field_op(zl.FR, zl.OP_TO_MONT, ks))
    rng = np.random.default_rng(2323)
    lo = rng.integers(0, 1 << 30, size=n, dtype=np.uint64)
    hi = rng.integers(0, 1 << 30, size=n, dtype=np.uint64)
    raw = np.zeros((n, 4), dtype=np.uint64)
    raw[:, 0] = lo
    raw[:, 2] = hi  # s_i = lo_i + 2^128 hi_i: low and high windows both in use
    sc = zl.field_op(zl.FR, zl.OP_TO_MONT, raw)
    idx = np.arange(1, n + 1, dtype=np.uint64)

    def closed(a, b, first):  # (sum (lo_i + 2^128 hi_i) * (first + i)) mod r over the given rows, exact in Python integers
        w = idx[:a.shape[0]] + np.uint64(first - 1)
        s_lo = sum(int(x) for x in (a * w).reshape(-1, 512).sum(axis=1)) if a.shape[0] % 512 == 0 else sum(int(x) * int(y) for x, y in zip(a, w))
        s_hi = sum(int(x) for x in (b * w).reshape(-1, 512).sum(axis=1)) if b.shape[0] % 512 == 0 else sum(int(x) * int(y) for x, y in zip(b, w))
        return (s_lo + (s_hi << 128)) % api.R_MOD

    b = zl.Bases.upload(bases, window_bits=17)
    try:
        assert b.plan()[0] == 17
        want = api.MSM.scalarMul(g, api.fr_from_int(closed(lo, hi, 1)))
        got = b.msm(sc)
        assert got[1] == want[1] and np.array_equal(got[0], want[0])
        m, off = 100352, 1000  # 196 * 512 points, far below two slices
        wsub = api.MSM.scalarMul(g, api.fr_from_int(closed(lo[:m], hi[:m], off + 1)))
        gsub = b.msm(sc[:m], off=off, n=m)
        assert gsub[1] == wsub[1] and np.array_equal(gsub[0], wsub[0])
    finally:
        b.free()


@pytest.mark.parametrize("span_pts", [700, 1300, 2600])
def test_device_scalar_path_in_point_slices(zl, ob, gm, span_pts, monkeypatch):
    """A launch set whose table rows would span more than ZG_MSM_TABLE_SPAN_MB is cut into slices of consecutive points
    (msm_enqueue_lane: every slice sorted and accumulated into its own set of bucket sums, the sets added up by
    msm_bucket_fold_kernel, ONE reduction). Same bytes as the oracle through every entry point that reaches it: synchronous,
    asynchronous (more calls in flight on two streams than the handle has workspaces), the ParallelMSM record mode, a
    sub-range, host scalars, infinity bases; 8, 4 and 2 slices (the last one shorter)."""
    import torch
    monkeypatch.setenv("ZG_MSM_TABLE_SPAN_MB", "1")
    monkeypatch.setenv("ZG_MSM_TABLE_SPAN_MIN_POINTS", str(span_pts))
    monkeypatch.setenv("ZG_MSM_HOST_SLICE_MIN", "1000")
    n = 5003
    inf = np.zeros(n, dtype=np.uint8)
    inf[7::40] = 1
    b = zl.Bases.upload(gm[:n], inf)
    dev = torch.device("cuda", 0)
    try:
        vecs = [ob.f_to_mont(ob.FR, U.random_raw256(4300 + j, n)) for j in range(5)]
        want = [ob.msm_g1(gm[:n], inf, v) for v in vecs]
        d = [torch.from_numpy(v.view(np.int64)).to(dev) for v in vecs]
        torch.cuda.synchronize()
        for j in range(2):
            got = b.msm_dev(d[j].data_ptr(), n)
            assert got[1] == want[j][1] and np.array_equal(got[0], want[j][0]), j
        streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
        outs = torch.zeros((20, 9), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        for rep in range(20):  # more calls in flight than workspaces
            j = rep % 5
            b.msm_dev_async(d[j].data_ptr(), n, outs[rep].data_ptr(), outs[rep, 8:].data_ptr(), stream=streams[rep % 2].cuda_stream)
        torch.cuda.synchronize()
        o = outs.cpu().numpy().view(np.uint64)
        for rep in range(20):
            j = rep % 5
            assert (o[rep, 8] & 0xFF) == want[j][1] and np.array_equal(o[rep, :8], want[j][0]), rep
        w2 = ob.msm_g1(gm[100:4100], inf[100:4100], vecs[0][:4000])
        g2 = b.msm_dev(d[0].data_ptr(), 4000, off=100)
        assert g2[1] == w2[1] and np.array_equal(g2[0], w2[0])
        jac = torch.zeros(12, dtype=torch.int64, device=dev)
        b.msm_partial_dev(d[1].data_ptr(), n, jac.data_ptr())
        torch.cuda.synchronize()
        rec = jac.cpu().numpy().view(np.uint64)
        assert np.array_equal(rec[:8], want[1][0]) and rec[8:].any()  # (x, y, 1): ParallelMSM's fromAffine record
        got = b.msm(vecs[2])  # host scalars
        assert got[1] == want[2][1] and np.array_equal(got[0], want[2][0])
        z = b.msm_dev(torch.zeros((n, 4), dtype=torch.int64, device=dev).data_ptr(), n)
        assert z[1] == 1
    finally:
        b.free()


def test_fixed_base_batch_equals_scalar_mul(zl, ob):
    """zg_g1_fixed_base_mul_batch (HyperKZG.setup's primitive) against the generic per-pair scalarMul kernel and the oracle:
    random scalars, 0, 1, r - 1, small values, single-window values, a base other than the generator, an infinity base."""
    from zolt_amd import api
    g = api.generator()
    n = 3000
    sc = ob.f_to_mont(ob.FR, U.random_raw256(4100, n))
    special = [0, 1, 2, 255, 256, 257, (1 << 248), (1 << 253), api.R_MOD - 1, api.R_MOD - 2, 0xFF << 64, 0x12345678]
    for i, v in enumerate(special):
        sc[i] = api.fr_from_int(v)
    out, inf = zl.g1_fixed_base_mul_batch(g, sc)
    w, wi = zl.g1_scalar_mul_batch(np.repeat(g[None, :], n, axis=0), np.zeros(n, dtype=np.uint8), sc)
    assert np.array_equal(inf, wi) and np.array_equal(out, w)
    assert inf[0] == 1 and not out[0].any()
    for i in list(range(len(special))) + [100, 2999]:
        o, oi = ob.g1_scalar_mul(g, 0, sc[i])
        assert inf[i] == oi and (oi or np.array_equal(out[i], o)), i
    base = w[50]  # some other point of the group
    out2, inf2 = zl.g1_fixed_base_mul_batch(base, sc[:64])
    w2, wi2 = zl.g1_scalar_mul_batch(np.repeat(base[None, :], 64, axis=0), np.zeros(64, dtype=np.uint8), sc[:64])
    assert np.array_equal(inf2, wi2) and np.array_equal(out2, w2)
    out3, inf3 = zl.g1_fixed_base_mul_batch(g, sc[:8], base_inf=1)
    assert inf3.all() and not out3.any()


def test_host_batch_of_long_vectors_interleaves_copies(zl, ob, gm, monkeypatch):
    """zg_msm_g1_batch with vectors too long to fuse: vector i's upload and launch set share stream i mod 3 (the copy of the next
    vector runs under the MSM of the previous one). Same results as k separate calls and as the oracle; k = 5 exercises the
    workspace rotation (3 lanes) and the join."""
    monkeypatch.setenv("ZG_MSM_HOST_SLICE_MIN", "1000")
    monkeypatch.setenv("ZG_MSM_BATCH_FUSE", "0")  # force the unfused path at a size the oracle finishes quickly
    n, k = 5003, 5
    b = zl.Bases.upload(gm[:n])
    try:
        batches = [ob.f_to_mont(ob.FR, U.random_raw256(4200 + j, n)) for j in range(k)]
        out, inf = b.msm_batch(batches)
        for j in range(k):
            w, wi = ob.msm_g1(gm[:n], None, batches[j])
            assert inf[j] == wi and np.array_equal(out[j], w), j
    finally:
        b.free()


@pytest.mark.parametrize("n", [1, 2, 63, 1000, 4097, 1 << 16])
def test_msm_over_machine_words_equals_msm_over_their_field_elements(zl, ob, n):
    """zg_msm_g1_u64 (the commit path of `zolt prove`: every committed polynomial is F.fromU64 of machine words, src/zkvm/mod.zig:
    1518-1617) == MSM.compute over the converted scalars, on the oracle and on the Montgomery entry point; edge words included"""
    gm = ob.g1_gen_multiples(n)
    rng = np.random.default_rng(n)
    v = rng.integers(0, 1 << 63, size=n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=n, dtype=np.uint64)
    v[:min(n, 5)] = np.array([0, 1, (1 << 64) - 1, 1 << 63, 255], dtype=np.uint64)[:min(n, 5)]
    sc = ob.f_to_mont(ob.FR, np.stack([v, np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.zeros(n, np.uint64)], axis=1))
    b = zl.Bases.upload(gm)
    try:
        want = ob.msm_g1(gm, None, sc)
        got = b.msm_u64(v)
        assert got[1] == want[1] and np.array_equal(got[0], want[0])
        again = b.msm(sc)
        assert again[1] == want[1] and np.array_equal(again[0], want[0])
        if n > 2:  # a range of the bases, as HyperKZG.commit of a shorter polynomial takes it
            w2 = ob.msm_g1(gm[1:n - 1], None, sc[:n - 2])
            g2 = b.msm_u64(v[:n - 2], n=n - 2, off=1)
            assert g2[1] == w2[1] and np.array_equal(g2[0], w2[0])
    finally:
        b.free()


@pytest.mark.parametrize("n", [0, 1, 2, 257, 70000])
def test_hyperkzg_setup_on_the_device(zl, ob, n):
    """zg_hyperkzg_setup: tau^i * G for i < n with the powers of tau, the fixed-base batch and the MSM handle all built in HBM == the
    oracle's generateMockSRS restatement (src/poly/commitment/mod.zig:174-213), across the three power tables (i < 256, < 65536, above);
    the handle it returns commits like an uploaded copy of the points, and works without the points ever coming back"""
    from zolt_amd import api
    g, tau = api.generator(), api.fr_from_int(api.HyperKZG.TAU)
    h, xy, inf = zl.Bases.hyperkzg_setup(g, tau, n)
    try:
        if n <= 300:
            want, winf = ob.hyperkzg_setup(n) if n else (np.zeros((0, 8), dtype=np.uint64), np.zeros(0, dtype=np.uint8))
            assert np.array_equal(xy, want) and not inf.any() and not np.asarray(winf).any()
        else:  # the oracle's double-and-add is slow: spot-check powers on both sides of the table boundaries with scalarMul
            for i in (0, 1, 255, 256, 257, 65535, 65536, 65537, n - 1):
                wxy, wi = api.MSM.scalarMul(g, api.fr_from_int(pow(api.HyperKZG.TAU, i, api.R_MOD)))
                assert wi == 0 and np.array_equal(xy[i], wxy), i
        if n:
            sc = ob.f_to_mont(ob.FR, U.random_raw256(77 + n, n))
            h2, _, _ = zl.Bases.hyperkzg_setup(g, tau, n, want_points=False)
            up = zl.Bases.upload(xy)
            a, b, c = h.msm(sc), h2.msm(sc), up.msm(sc)
            assert a[1] == b[1] == c[1] and np.array_equal(a[0], b[0]) and np.array_equal(a[0], c[0])
            h2.free()
            up.free()
    finally:
        h.free()


def test_hyperkzg_setup_with_tau_zero_keeps_the_identity_flags(zl, ob):
    """tau = 0 (the ABI accepts any field element): powers[0] = G, every later power is scalarMul(g1, 0) = the identity, as in the
    reference's loop (src/poly/commitment/mod.zig:194-199). The handle must carry those flags — a commitment over it is evals[0] * G and
    nothing else (round-5 advisor: the flags were dropped and (0, 0) entered the MSM as a point)."""
    from zolt_amd import api
    g, n = api.generator(), 300
    h, xy, inf = zl.Bases.hyperkzg_setup(g, np.zeros(4, dtype=np.uint64), n)
    try:
        assert inf[0] == 0 and np.array_equal(xy[0], g) and inf[1:].all() and not xy[1:].any()
        sc = ob.f_to_mont(ob.FR, U.random_raw256(4040, n))
        got = h.msm(sc)
        want = ob.msm_g1(xy, inf, sc)
        w1 = api.MSM.scalarMul(g, sc[0])
        assert got[1] == want[1] == w1[1] == 0 and np.array_equal(got[0], want[0]) and np.array_equal(got[0], w1[0])
    finally:
        h.free()
