"""Shared helpers for the tests: representation conversions and seeded inputs.

Conversions go through oracle/pymodel.py big-ints (test infrastructure)."""
import json
import os

import numpy as np

from oracle import pymodel as pm

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
M64 = (1 << 64) - 1


def load_vectors():
    with open(os.path.join(GOLDEN, "vectors.json")) as f:
        return json.load(f)


def mont_limbs(v, mod):
    """canonical int -> (4,) uint64 Montgomery limbs"""
    return np.array(pm.limbs(pm.to_mont(v, mod)), dtype=np.uint64)


def fr(vals):
    return np.array([pm.limbs(pm.to_mont(int(v), pm.R_MOD)) for v in vals], dtype=np.uint64).reshape(-1, 4)


def fp(vals):
    return np.array([pm.limbs(pm.to_mont(int(v), pm.P_MOD)) for v in vals], dtype=np.uint64).reshape(-1, 4)


def fr_hex(hexes):
    return fr([int(h, 16) for h in hexes])


def fr_to_int(l):
    return pm.from_mont(pm.from_limbs(l), pm.R_MOD)


def fp_to_int(l):
    return pm.from_mont(pm.from_limbs(l), pm.P_MOD)


def points_xy(pts):
    """list of (x,y) canonical ints -> (n,8) uint64 Montgomery"""
    out = np.zeros((len(pts), 8), dtype=np.uint64)
    for i, p in enumerate(pts):
        if p is None:
            continue
        out[i, :4] = pm.limbs(pm.to_mont(p[0], pm.P_MOD))
        out[i, 4:] = pm.limbs(pm.to_mont(p[1], pm.P_MOD))
    return out


def point_from_xy(xy, inf):
    if inf:
        return None
    return (fp_to_int(xy[:4]), fp_to_int(xy[4:]))


def splitmix64(seed, n):
    """n uint64 words of splitmix64(seed) (SURVEY §8(d) synthetic inputs) — vectorised."""
    idx = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def random_raw256(seed, n):
    """(n,4) uint64 raw 256-bit integers (may exceed the modulus)."""
    return splitmix64(seed, 4 * n).reshape(n, 4)
