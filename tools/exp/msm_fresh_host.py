#!/usr/bin/env python3
"""zg_msm_g1 (host scalars) as an UNMODIFIED call site reaches it: every call hands over a freshly allocated scalar vector at a NEW address
(the previous ones stay alive, as a prover's polynomials do) — against the same call on one long-lived vector (what bench.py times)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from zolt_amd import lib, api
from bench import raw_scalars
lib.init(0)
v = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << v
h, _, _ = lib.Bases.hyperkzg_setup(api.generator(), api.fr_from_int(0x12345678), n, want_points=False)
sc = lib.field_op(lib.FR, lib.OP_TO_MONT, raw_scalars(0x534D414C, 0, n))
for _ in range(3): h.msm(sc)
t = []
for _ in range(8):
    t0 = time.perf_counter(); h.msm(sc); t.append((time.perf_counter() - t0) * 1e3)
print("long-lived vector      ms", [round(x, 2) for x in t])
keep, t = [], []
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 8):
    a = sc.copy()  # a new allocation at a new address (the old ones are kept)
    keep.append(a)
    t0 = time.perf_counter(); h.msm(a); t.append((time.perf_counter() - t0) * 1e3)
print("fresh vector per call  ms", [round(x, 2) for x in t], "mean", round(float(np.mean(t)), 2), "median", round(float(np.median(t)), 2))
t = []
for a in keep:
    t0 = time.perf_counter(); h.msm(a); t.append((time.perf_counter() - t0) * 1e3)
print("the same vectors again ms", [round(x, 2) for x in t])
