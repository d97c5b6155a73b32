#!/usr/bin/env python3
"""Extracts what the reference's captured run printed about Stage 1 — the UniSkip first round and the nine remaining rounds of its
StreamingOuterProver (Spartan outer sumcheck, 256 cycles) — into tests/golden/stage1_outer_rounds.json. Data only, no source text.

Source: /root/reference/logs/zolt.log:1572-2091, printed by src/zkvm/proof_converter.zig:380-540, src/zkvm/spartan/streaming_outer.zig
and src/poly/split_eq.zig:429-431:
  tau[0..10)             the ten 125-bit challenges as stored limbs [0, 0, lo, hi] (raw Montgomery form, as the C ABI takes them)
  uni_poly_coeffs[0..2)  the two coefficients of the UniSkip polynomial the log prints (canonical, big-endian)
  r0, w[0..10)           the first-round challenge and the Lagrange basis values L_i(r0) over {-4..5} (canonical, big-endian)
  lagrange_tau_r0        L(r0, tau_high) as raw limbs — the split-eq structure's initial scalar
  uni_skip_claim, batching_coeff (big-endian), the initial batched claim (little-endian)
  rounds[0..9)           Gruen's q(0), q(1) and previous_claim (big-endian; `index` = split_eq.current_index), the batched compressed
                         coefficients c0, c2, c3 and the challenge (little-endian)
  final                  split_eq.current_scalar after the nine binds (little-endian and raw limbs)
The cycle witnesses are not in the log, so the sums t'(0), t'(inf) themselves are held against the restatement only; what this fixture
pins is everything around them: Lagrange weights, the kernel, the claim chain through Gruen's cubic, the scalar's binds.

Run in the build container (needs /root/reference); the JSON it writes is committed."""
import json
import os
import re

LOG = "/root/reference/logs/zolt.log"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stage1_outer_rounds.json")


def braces(line):
    return [bytes(int(x) for x in g.replace(" ", "").strip(",").split(",")).hex() for g in re.findall(r"\{ ?([0-9, ]+?) ?\}", line)]


def limbs(line):
    return [int(x, 16) for x in re.search(r"\[([0-9a-fx, ]+)\]\s*$", line).group(1).replace("0x", "").split(",")]


def main():
    lines = open(LOG, errors="replace").read().splitlines()
    out = {"source": "logs/zolt.log:1572-2091", "tau_limbs": [], "uni_poly_coeffs_be": [], "w_be": [], "rounds": []}
    cur = None
    for i, l in enumerate(lines[:2100]):
        if re.match(r"\[ZOLT PROVE\] tau\[\d+\] = challengeScalar", l):
            j = next(k for k in range(i, i + 10) if "result_limbs" in lines[k])
            out["tau_limbs"].append(limbs(lines[j]))
        m = re.match(r"\[ZOLT UNISKIP_PROOF\] uni_poly_coeffs\[(\d+)\] = ", l)
        if m and len(out["uni_poly_coeffs_be"]) == int(m.group(1)):
            out["uni_poly_coeffs_be"].append(braces(l)[0])
        if l.startswith("[STREAMING_OUTER] lagrange_tau_r0 (limbs)"):
            out["lagrange_tau_r0_limbs"] = limbs(l)
        if l.startswith("[ZOLT] STAGE1: uni_skip_claim@SpartanOuter"):
            out["uni_skip_claim_be"] = braces(l)[0]
        if l.startswith("[ZOLT] computeLagrangeEvalsAtR0: r0 = "):
            out["r0_be"] = braces(l)[0]
        m = re.match(r"\[ZOLT\] computeLagrangeEvalsAtR0: w\[(\d+)\] = ", l)
        if m:
            out["w_be"].append(braces(l)[0])
        if l.startswith("[ZOLT] STAGE1: batching_coeff = "):
            out["batching_coeff_be"] = braces(l)[0]
        if l.startswith("[ZOLT] STAGE1_INITIAL: claim = "):
            out["initial_claim_le"] = braces(l)[0]
        m = re.match(r"\[GRUEN ROUND (\d+)\] q\(0\) = ", l)
        if m and "batching_coeff_be" in out and len(out["rounds"]) < 9:
            cur = {"index": int(m.group(1)), "q0_be": braces(l)[0]}
            out["rounds"].append(cur)
        elif cur is not None:
            for key, pat in (("q1_be", r"\[GRUEN ROUND \d+\] q\(1\) = "), ("previous_claim_be", r"\[GRUEN ROUND \d+\] previous_claim = "),
                             ("c0_le", r"\[ZOLT\] STAGE1_ROUND_\d+: c0 = "), ("c2_le", r"\[ZOLT\] STAGE1_ROUND_\d+: c2 = "),
                             ("c3_le", r"\[ZOLT\] STAGE1_ROUND_\d+: c3 = "), ("challenge_le", r"\[ZOLT\] STAGE1_ROUND_\d+: challenge = ")):
                if re.match(pat, l) and key not in cur:
                    cur[key] = braces(l)[0]
        if l.startswith("[ZOLT] STAGE1_FINAL: prover eq_factor (split_eq.current_scalar) = "):
            out["final_eq_factor_le"] = braces(l)[0]
        if l.startswith("[ZOLT] STAGE1_FINAL: prover eq_factor limbs = "):
            out["final_eq_factor_limbs"] = limbs(l)
    assert len(out["tau_limbs"]) == 10 and len(out["w_be"]) == 10 and len(out["uni_poly_coeffs_be"]) == 2
    assert len(out["rounds"]) == 9 and all(len(r) == 8 for r in out["rounds"]), out["rounds"]
    assert [r["index"] for r in out["rounds"]] == list(range(9, 0, -1))
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
