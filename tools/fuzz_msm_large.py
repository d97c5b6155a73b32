#!/usr/bin/env python3
"""Randomised MSMs at sizes the CPU oracle cannot follow (2^17 .. 2^21 points), checked against the closed form: bases (i+1) G, so
MSM(bases[off:off+m], s) = (sum s_j (off + j + 1) mod r) G, one scalarMul through an independent kernel. Random zg_msm_config
(window_bits, precompute_levels, expected_uses), random sub-ranges, three scalar families (uniform mod r, machine words through
zg_msm_g1_u64, 0/1 columns), host and resident scalars, and the ambient code-path switches of tools/fuzz_msm.py. The size thresholds of the
plan (17-bit windows, the two-pass sort's fine bits, point slices, heavy buckets) all lie in this range.
usage: fuzz_msm_large.py [seconds=120] [seed=1]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401  (before the library: it bundles its own HIP runtime)
from zolt_amd import api, lib  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
lib.init(0)
NMAX = 1 << 21
g = api.generator()
ks = np.zeros((NMAX, 4), dtype=np.uint64)
ks[:, 0] = np.arange(1, NMAX + 1, dtype=np.uint64)
GM, _ = lib.g1_fixed_base_mul_batch(g, lib.field_op(lib.FR, lib.OP_TO_MONT, ks))
del ks

AMBIENT = {"ZG_MSM_TWO_PASS_SORT": ["0"], "ZG_MSM_LDS_SORT": ["0"], "ZG_MSM_REDUCE_2D": ["0"], "ZG_MSM_ALONE_FULL": ["0"], "ZG_MSM_SIDE_TABLE": ["0"],
           "ZG_MSM_FINE_BITS": ["5", "6"], "ZG_MSM_FINE_BITS_MIN": ["7", "8"], "ZG_MSM_HOST_AFFINE": ["0"], "ZG_MSM_ROWCOL_WAVE_FROM": ["0", "1"],
           "ZG_MSM_TABLE_SPAN_MB": ["16", "64", "256"], "ZG_MSM_LANES": ["1", "2"], "ZG_MSM_SLICE_LOCAL_REFS": ["0"], "ZG_MSM_HOST_SLICES": ["1", "2", "5"],
           "ZG_MSM_C17_MIN": ["100000", "3000000"]}


def draw_ambient():
    for k in AMBIENT:
        os.environ.pop(k, None)
    picked = {}
    if rng.random() < 0.5:
        for k in rng.choice(sorted(AMBIENT), size=int(rng.choice([1, 2, 3])), replace=False):
            picked[str(k)] = str(rng.choice(AMBIENT[str(k)]))
            os.environ[str(k)] = picked[str(k)]
    return picked


def weighted(raw_words, first):
    """sum_j value_j * (first + j) for values given as (m, L) u64 limbs, exact (32-bit halves, rows of 64)"""
    m = raw_words.shape[0]
    pad = (-m) % 64
    w = np.arange(first, first + m + pad, dtype=np.uint64)
    tot = 0
    for limb in range(raw_words.shape[1]):
        col = np.concatenate([raw_words[:, limb], np.zeros(pad, dtype=np.uint64)])
        for half, shift in ((col & np.uint64(0xFFFFFFFF), 0), (col >> np.uint64(32), 32)):
            tot += sum((half * w).reshape(-1, 64).sum(axis=1, dtype=np.uint64).tolist()) << (64 * limb + shift)
    return tot


t0, cases, refused = time.time(), 0, 0
while time.time() - t0 < budget:
    ambient = draw_ambient()
    n = int(rng.choice([1 << 17, (1 << 17) + 1, 200000, 1 << 18, 300001, 1 << 19, 900001, 1 << 20, (1 << 20) + 1, 1500000, 1 << 21]))
    cfg = {"window_bits": int(rng.choice([0, 0, 0, 8, 10, 12, 13, 14, 15, 16, 17, 18, 19])), "precompute_levels": int(rng.choice([0, 0, 0, 1, 2, 3, 5, 8, 16, 20])),
           "expected_uses": int(rng.choice([0, 0, 1, 3]))}
    try:
        b = lib.Bases.upload(GM[:n], None, **cfg)
    except lib.ZgError:
        refused += 1  # a documented refusal (too many buckets / groups for this window and level count), not a mismatch
        continue
    try:
        for _ in range(2):
            off = int(rng.integers(0, n // 2)) if rng.random() < 0.4 else 0
            m = int(rng.integers(1, n - off + 1)) if rng.random() < 0.4 else n - off
            kind = int(rng.integers(0, 4))
            if kind == 1:  # machine words
                words = rng.integers(0, 1 << 63, size=m, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=m, dtype=np.uint64)
                words[rng.random(m) < 0.3] = 0
                got = b.msm_u64(words, n=m, off=off)
                k = weighted(words[:, None], off + 1) % api.R_MOD
            else:
                raw = rng.integers(0, 1 << 63, size=(m, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(m, 4), dtype=np.uint64)
                if kind == 2:  # a 0/1 column: half of all points in one bucket
                    raw[:] = 0
                    raw[:, 0] = rng.integers(0, 2, size=m, dtype=np.uint64)
                raw[:, 3] &= np.uint64((1 << 61) - 1)  # below r: the scalar is the raw value
                sc = lib.field_op(lib.FR, lib.OP_TO_MONT, raw)
                if kind == 3:
                    d = torch.from_numpy(sc.view(np.int64)).cuda()
                    torch.cuda.synchronize()
                    got = b.msm_dev(d.data_ptr(), m, off=off)
                else:
                    got = b.msm(sc, off=off, n=m)
                k = weighted(raw, off + 1) % api.R_MOD
            want = api.MSM.scalarMul(g, api.fr_from_int(k))
            assert got[1] == want[1] and np.array_equal(got[0], want[0]), ("msm", n, cfg, off, m, kind, ambient, b.plan())
            cases += 1
    finally:
        b.free()
print(f"fuzz ok: {cases} large MSMs checked against the closed form in {time.time() - t0:.1f} s ({refused} configurations refused at upload)")
