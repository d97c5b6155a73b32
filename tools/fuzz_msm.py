#!/usr/bin/env python3
"""Randomised differential test of the MSM entry points against the CPU oracle (not part of the pytest suite: run it for as
long as you like).  usage: fuzz_msm.py [seconds=60] [seed=1]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import binding as ob  # checker
from zolt_amd import lib

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
lib.init(0)
lib.init_devices(1)
NMAX = 70000
gm = ob.g1_gen_multiples(NMAX)
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def scalars(n, kind):
    if kind == 0:
        raw = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
        return ob.f_to_mont(ob.FR, raw)
    if kind == 1:
        return ob.f_from_u64(ob.FR, rng.integers(0, 2, size=n).astype(np.uint64))
    if kind == 2:
        return ob.f_from_u64(ob.FR, rng.integers(0, 1 << 16, size=n).astype(np.uint64))
    if kind == 3:
        return ob.f_from_u64(ob.FR, np.full(n, int(rng.integers(1, 1 << 40)), dtype=np.uint64))
    sc = scalars(n, 0)
    if n == 0:
        return sc
    k = max(1, n // 7)
    idx = rng.integers(0, n, size=k)
    vals = [0, 1, R - 1, R - 2, (1 << 15), (1 << 16) - 1, (1 << 16), (1 << 16) + 1, (1 << 17) - 1, (1 << 17), (1 << 255) % R, ((1 << 240) - 1),
            (1 << 238) - 1, (1 << 253) + (1 << 16)]
    raw = np.array([[(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)] for v in vals], dtype=np.uint64)
    sc[idx] = ob.f_to_mont(ob.FR, raw)[rng.integers(0, len(vals), size=k)]
    return sc


# ambient code-path switches, drawn per iteration before the handle is made (see tools/fuzz_open_rwc.py: round 6)
AMBIENT = {"ZG_MSM_TWO_PASS_SORT": ["0"], "ZG_MSM_LDS_SORT": ["0"], "ZG_MSM_REDUCE_2D": ["0"], "ZG_MSM_ALONE_FULL": ["0"], "ZG_MSM_BATCH_FUSE": ["0"],
           "ZG_MSM_SIDE_TABLE": ["0"], "ZG_MSM_FINE_BITS": ["5", "6"], "ZG_MSM_FINE_BITS_MIN": ["7", "8"], "ZG_MSM_HOST_AFFINE": ["0"],
           "ZG_MSM_ROWCOL_WAVE_FROM": ["0", "1"], "ZG_MSM_LANES": ["1", "2"], "ZG_MSM_CHUNK_SCHED": ["0"], "ZG_MSM_COMBINE_PER_QUAD": ["8"],
           "ZG_MSM_SLICE_LOCAL_REFS": ["0"]}


def draw_ambient():
    for k in AMBIENT:
        os.environ.pop(k, None)
    picked = {}
    if rng.random() < 0.4:
        for k in rng.choice(sorted(AMBIENT), size=int(rng.choice([1, 1, 2, 3])), replace=False):
            picked[str(k)] = str(rng.choice(AMBIENT[str(k)]))
            os.environ[str(k)] = picked[str(k)]
    return picked


t0 = time.time()
cases = 0
while time.time() - t0 < budget:
    ambient = draw_ambient()
    big = rng.random() < 0.25
    n = int(rng.integers(32768, NMAX)) if big else int(rng.choice([0, 1, 2, 7, 8, 9, 31, 33, 100, 511, 1000, 2048, 4097, 9000, 20000]))
    perm_dup = rng.random() < 0.3
    xy = gm[:n].copy()
    if perm_dup and n > 4:  # repeated and negated points in the same bucket sets
        xy[n // 2:] = xy[:n - n // 2]
        if rng.random() < 0.5:
            neg = xy[n // 2:].copy()
            neg[:, 4:] = ob.f_neg(ob.FP, neg[:, 4:])
            xy[n // 2:] = neg
    inf = (rng.random(n) < 0.05).astype(np.uint8) if rng.random() < 0.5 else None
    cfg = {}
    if rng.random() < 0.5:
        cfg["window_bits"] = int(rng.choice([2, 3, 5, 7, 8, 10, 11, 12, 13, 14, 15, 16, 17]))
    if rng.random() < 0.4:
        cfg["precompute_levels"] = int(rng.integers(1, 20))
    try:
        b = lib.Bases.upload(xy, inf, **cfg)
    except lib.ZgError as e:  # e.g. too many groups for a tiny window with few levels: a documented refusal, not a mismatch
        continue
    for _ in range(2):
        sc = scalars(n, int(rng.integers(0, 5)))
        off = int(rng.integers(0, n + 1)) if rng.random() < 0.3 else 0
        m = int(rng.integers(0, n - off + 1)) if off or rng.random() < 0.3 else n
        got, ginf = b.msm(sc[:m], off=off, n=m)
        want, winf = ob.msm_g1(xy[off:off + m], None if inf is None else inf[off:off + m], sc[:m])
        assert ginf == winf and np.array_equal(got, want), ("msm", n, cfg, off, m, ambient)
        cases += 1
    if n and rng.random() < 0.5:
        k = int(rng.integers(2, 12))
        nb = int(rng.integers(1, min(n, 6000) + 1))
        batches = [scalars(nb, int(rng.integers(0, 5))) for _ in range(k)]
        outs, infs = b.msm_batch(batches, n=nb)
        for j in range(k):
            want, winf = ob.msm_g1(xy[:nb], None if inf is None else inf[:nb], batches[j])
            assert infs[j] == winf and np.array_equal(outs[j], want), ("batch", n, cfg, nb, k, j, ambient)
        cases += k
    b.free()
    # round-2 entry points: the one-process multi-GPU path with a random number of logical shards (peer-copy exchange on this box),
    # the sliced host-scalar path, the fixed-base batch and the affine group law
    if n and rng.random() < 0.5:
        os.environ["ZG_SHARDS"] = str(int(rng.integers(1, 9)))
        os.environ["ZG_MSM_HOST_SLICES"] = str(int(rng.integers(1, 9)))
        os.environ["ZG_MSM_HOST_SLICE_MIN"] = str(int(rng.integers(1, 5000)))
        sb = lib.ShardedBases.upload(xy, inf)
        sc = scalars(n, int(rng.integers(0, 5)))
        m = int(rng.integers(0, n + 1)) if rng.random() < 0.4 else n
        got, ginf = sb.msm(sc[:m], m)
        want, winf = ob.msm_g1(xy[:m], None if inf is None else inf[:m], sc[:m])
        assert ginf == winf and np.array_equal(got, want), ("sharded", n, m, os.environ["ZG_SHARDS"])
        k = int(rng.integers(1, 5))
        nb = int(rng.integers(0, min(n, 4000) + 1))
        batches = [scalars(nb, int(rng.integers(0, 5))) for _ in range(k)]
        outs, infs = sb.msm_batch(batches, nb)
        for j in range(k):
            want, winf = ob.msm_g1(xy[:nb], None if inf is None else inf[:nb], batches[j])
            assert infs[j] == winf and np.array_equal(outs[j], want), ("sharded batch", n, nb, k, j)
        sb.free()
        b2 = lib.Bases.upload(xy, inf)
        got, ginf = b2.msm(sc)  # host scalars: sliced when n >= the random minimum
        want, winf = ob.msm_g1(xy, inf, sc)
        assert ginf == winf and np.array_equal(got, want), ("sliced host path", n, os.environ["ZG_MSM_HOST_SLICES"])
        b2.free()
        cases += 2 + k
    if rng.random() < 0.3:
        m = int(rng.integers(1, 300))
        sc = scalars(m, int(rng.integers(0, 5)))
        base = gm[int(rng.integers(0, NMAX))]
        fx, fi = lib.g1_fixed_base_mul_batch(base, sc)
        gx, gi = lib.g1_scalar_mul_batch(np.repeat(base[None, :], m, axis=0), np.zeros(m, dtype=np.uint8), sc)
        assert np.array_equal(fi, gi) and np.array_equal(fx, gx), "fixed base"
        for j in range(min(m, 3)):
            o, oi = ob.g1_scalar_mul(base, 0, sc[j])
            assert fi[j] == oi and (oi or np.array_equal(fx[j], o))
        a_i, b_i = rng.integers(0, NMAX, size=m), rng.integers(0, NMAX, size=m)
        b_i[: m // 4] = a_i[: m // 4]  # P + P
        pa, pb = gm[a_i].copy(), gm[b_i].copy()
        neg = slice(m // 4, m // 2)
        pb[neg] = pa[neg]
        pb[neg, 4:] = ob.f_neg(ob.FP, pa[neg, 4:])  # P + (-P)
        ia, ib = (rng.random(m) < 0.1).astype(np.uint8), (rng.random(m) < 0.1).astype(np.uint8)
        ox, oi = lib.g1_affine_add_batch(pa, ia, pb, ib)
        for j in range(m):
            w, wi = ob.g1_add_affine(pa[j], int(ia[j]), pb[j], int(ib[j]))
            assert oi[j] == wi and (wi or np.array_equal(ox[j], w)), ("affine add", j)
        cases += 2
    # round-5 entry points: handles planned for a few uses (no table; batches of rows share one reduction), MSMs over machine words,
    # the proving key built on the device with a random fixed-base window width
    if n and rng.random() < 0.4:
        os.environ["ZG_MSM_ROWS_SHARED_TAIL"] = str(int(rng.integers(0, 2)))
        bt = lib.Bases.upload(xy, inf, expected_uses=int(rng.integers(1, 16)))
        k = int(rng.integers(2, 9))
        nb = int(rng.integers(1, n + 1))
        batches = [scalars(nb, int(rng.integers(0, 5))) for _ in range(k)]
        outs, infs = bt.msm_batch(batches, n=nb)
        for j in range(k):
            want, winf = ob.msm_g1(xy[:nb], None if inf is None else inf[:nb], batches[j])
            assert infs[j] == winf and np.array_equal(outs[j], want), ("table-less batch", n, nb, k, j, os.environ["ZG_MSM_ROWS_SHARED_TAIL"])
        words = rng.integers(0, 1 << 63, size=nb, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=nb, dtype=np.uint64)
        if rng.random() < 0.5:
            words[rng.random(nb) < 0.5] = 0
        got, ginf = bt.msm_u64(words, n=nb)
        want, winf = ob.msm_g1(xy[:nb], None if inf is None else inf[:nb], ob.f_from_u64(ob.FR, words))
        assert ginf == winf and np.array_equal(got, want), ("u64 msm", n, nb)
        bt.free()
        cases += k + 1
    if rng.random() < 0.15:
        from zolt_amd import api
        m = int(rng.choice([1, 2, 33, 1000, 20000, 40000]))
        os.environ["ZG_FB_WINDOW_BITS"] = str(int(rng.choice([0, 4, 7, 8, 10, 11, 12, 13])))
        tau = scalars(1, 0)[0]
        h, pxy, pinf = lib.Bases.hyperkzg_setup(gm[0], tau, m, expected_uses=int(rng.integers(0, 3)))
        pw = [1]
        for _ in range(m - 1):
            pw.append(pw[-1] * api.fr_to_int(tau) % R)
        for j in (0, m // 2, m - 1):
            o, oi = ob.g1_scalar_mul(gm[0], 0, api.fr_from_int(pw[j]))
            assert pinf[j] == oi and (oi or np.array_equal(pxy[j], o)), ("setup power", m, j)
        sc = scalars(m, int(rng.integers(0, 5)))
        got, ginf = h.msm(sc)
        want, winf = ob.msm_g1(pxy, pinf, sc)
        assert ginf == winf and np.array_equal(got, want), ("msm over the device-built key", m)
        h.free()
        os.environ.pop("ZG_FB_WINDOW_BITS")
        cases += 1
print(f"fuzz ok: {cases} MSMs checked in {time.time() - t0:.1f} s")
