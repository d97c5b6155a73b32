#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/.

Run in the build container (needs /root/reference for the two data files):
    python tests/golden/make_golden.py

Outputs (all DATA — inputs and expected outputs, no reference source text):
  fibonacci.elf             copy of the reference's example guest binary
                            (/root/reference/examples/fibonacci.elf, 4872 B) — MSM input
  zolt_proof_regular.bin    copy of the reference's captured ZOLT-v1 proof
                            (/root/reference/logs/zolt_proof_regular.bin, 11345 B) —
                            holds real HyperKZG.commit outputs of the reference
  stage1_tau.json           13 tau challenges of the captured run (logs/zolt.log:47-59)
  vectors.json              expected outputs computed by the independent Python
                            big-int model oracle/pymodel.py (NOT by the C oracle, NOT
                            by the GPU code) on seeded inputs
"""
import json
import os
import random
import re
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import pymodel as pm  # noqa: E402

REF = "/root/reference"


def hx(v):
    return "%064x" % v


def main():
    shutil.copyfile(f"{REF}/examples/fibonacci.elf", f"{HERE}/fibonacci.elf")
    shutil.copyfile(f"{REF}/logs/zolt_proof_regular.bin", f"{HERE}/zolt_proof_regular.bin")
    os.chmod(f"{HERE}/fibonacci.elf", 0o644)
    os.chmod(f"{HERE}/zolt_proof_regular.bin", 0o644)

    taus = []
    with open(f"{REF}/logs/zolt.log") as f:
        for line in f:
            m = re.match(r"\[PROVER STAGE 1\]\s+tau\[(\d+)\] = ([0-9a-f]{64})", line)
            if m:
                taus.append(m.group(2))
    assert len(taus) == 13
    json.dump({"source": "logs/zolt.log:47-59", "tau_hex": taus}, open(f"{HERE}/stage1_tau.json", "w"), indent=1)

    rng = random.Random(0x5A4F4C54)
    vec = {"seed": "0x5A4F4C54", "format": "canonical (non-Montgomery) big-endian hex; point = [x,y] or null"}

    # --- MSM vectors: bases k_i*G with random k_i, uniform scalars, edge cases mixed in
    msm_cases = []
    for n in (1, 2, 7, 8, 9, 31, 32, 33, 100):
        ks = [rng.randrange(1, pm.R_MOD) for _ in range(n)]
        sc = [rng.randrange(0, pm.R_MOD) for _ in range(n)]
        if n >= 8:
            sc[0] = 0
            sc[1] = 1
            sc[2] = pm.R_MOD - 1
            ks[4] = ks[3]                      # duplicate point
            ks[5] = pm.R_MOD - ks[3]           # its negation
            sc[5] = sc[3]                      # P*s + (-P)*s cancels
        pts = [pm.ec_mul(k, pm.G1) for k in ks]
        infs = [0] * n
        if n >= 9:
            infs[6] = 1                        # an infinity base (skipped, src/msm/mod.zig:407)
        res = pm.msm([None if i else p for p, i in zip(pts, infs)], sc)
        msm_cases.append({
            "n": n,
            "points": [[hx(p[0]), hx(p[1])] for p in pts],
            "inf": infs,
            "scalars": [hx(s) for s in sc],
            "result": None if res is None else [hx(res[0]), hx(res[1])],
        })
    vec["msm"] = msm_cases

    # --- bench family: P_i = (i+1)G, s_i = 7i+13 (src/bench.zig:261-268), closed form
    bench = []
    for n in (16, 64, 256, 1000):
        ks = list(range(1, n + 1))
        sc = [7 * i + 13 for i in range(n)]
        res = pm.msm_generator_multiples(ks, sc)
        bench.append({"n": n, "result": [hx(res[0]), hx(res[1])]})
    vec["msm_bench_family"] = bench
    vec["generator_multiples"] = [[hx(c) for c in pm.ec_mul(k, pm.G1)] for k in range(1, 9)]

    # --- mock SRS (tau = 0x12345678) first points, src/poly/commitment/mod.zig:189-199
    srs = pm.mock_srs(6)
    vec["mock_srs"] = [[hx(p[0]), hx(p[1])] for p in srs]

    # --- eq tables, folds, sumcheck
    eqs = []
    for v in (0, 1, 2, 3, 5):
        r = [rng.randrange(0, pm.R_MOD) for _ in range(v)]
        eqs.append({"r": [hx(x) for x in r], "table": [hx(x) for x in pm.eq_table(r)]})
    vec["eq_table"] = eqs
    folds = []
    for v in (1, 3, 6):
        t = [rng.randrange(0, pm.R_MOD) for _ in range(1 << v)]
        r = rng.randrange(0, pm.R_MOD)
        folds.append({"table": [hx(x) for x in t], "r": hx(r),
                      "bind_high": [hx(x) for x in pm.bind_high(t, r)],
                      "bind_low": [hx(x) for x in pm.bind_low(t, r)]})
    vec["folds"] = folds
    scs = []
    for v in (1, 3, 6):
        t = [rng.randrange(0, pm.R_MOD) for _ in range(1 << v)]
        claim, rounds, chals, fin, ok = pm.run_sumcheck(t)
        scs.append({"evals": [hx(x) for x in t], "claim": hx(claim),
                    "rounds": [[hx(c) for c in rd] for rd in rounds],
                    "challenges": [hx(c) for c in chals], "final_eval": hx(fin), "ok": ok})
    vec["sumcheck"] = scs
    json.dump(vec, open(f"{HERE}/vectors.json", "w"))
    print("wrote fixtures to", HERE)


if __name__ == "__main__":
    main()
