#!/usr/bin/env python3
"""Extracts the LassoProver run the reference captured in its own log (Stage 3: log_K = 16 address rounds + log_T = 8 cycle rounds over
44 lookups) into tests/golden/lasso_rounds.json — data only: numbers the reference printed, no source text.

Source: /root/reference/logs/zolt.log:430-970, printed by LassoProver.computeRoundPolynomial / receiveChallenge
(src/zkvm/lasso/prover.zig:262-453) as the RAW limbs of each field element, limbs[3]..limbs[0] — i.e. the Montgomery
representative, exactly the uint64[4] format of the C ABI:
  per round   phase, current_claim, the round polynomial's coefficients c[0..2] = [sum_0, sum_1 - sum_0, 0], p(0), p(1),
              the challenge the transcript produced, and current_claim after the bind (the sum of the updated eq_evals)
The 8 r_reduction challenges are logged only by their low 64 bits, so the prover's input tables cannot be rebuilt; what the file pins
is every relation between the printed numbers: p(0) + p(1) = claim, claim_after = p(challenge), claim_after = the next round's claim
(see tests/test_transcript_host.py::test_lasso_rounds_of_the_captured_run and the GPU twin in tests/test_gpu_prover_sites.py).

Run in the build container (needs /root/reference); the JSON it writes is committed."""
import json
import os
import re

LOG = "/root/reference/logs/zolt.log"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lasso_rounds.json")


def main():
    lines = open(LOG, errors="replace").read().splitlines()
    rounds, cur = [], None
    meta = {}
    for l in lines:
        m = re.match(r"\[PROVER STAGE 3\]   log_K=(\d+), log_T=(\d+), total_rounds=(\d+)", l)
        if m:
            meta = {"log_K": int(m.group(1)), "log_T": int(m.group(2)), "total_rounds": int(m.group(3))}
        m = re.match(r"\[PROVER STAGE 3\]   num_lookup_entries=(\d+)", l)
        if m:
            meta["num_lookup_entries"] = int(m.group(1))
        m = re.match(r"\[LASSO PROVER\] computeRoundPolynomial round=(\d+) phase=(\w+)", l)
        if m:
            cur = {"round": int(m.group(1)), "phase": m.group(2)}
            rounds.append(cur)
            continue
        if cur is None:
            continue
        for key, pat in (("claim", r"\[LASSO PROVER\] current_claim = ([0-9a-f]{64})$"), ("c0", r"\[LASSO PROVER\]   c\[0\] = ([0-9a-f]{64})"),
                         ("c1", r"\[LASSO PROVER\]   c\[1\] = ([0-9a-f]{64})"), ("c2", r"\[LASSO PROVER\]   c\[2\] = ([0-9a-f]{64})"),
                         ("p0", r"\[LASSO PROVER\] p\(0\) = ([0-9a-f]{64})"), ("p1", r"\[LASSO PROVER\] p\(1\) = ([0-9a-f]{64})"),
                         ("challenge", r"\[LASSO PROVER\] challenge = ([0-9a-f]{64})"),
                         ("claim_after", r"\[LASSO PROVER\] current_claim \(after\) = ([0-9a-f]{64})")):
            m = re.match(pat, l)
            if m and key not in cur:
                cur[key] = m.group(1)  # 64 hex digits = limbs[3] limbs[2] limbs[1] limbs[0] of the Montgomery representative
    rounds = [r for r in rounds if "challenge" in r and "claim_after" in r]
    assert len(rounds) == meta["total_rounds"] == 24 and [r["round"] for r in rounds] == list(range(24))
    out = dict(meta, encoding="64 hex digits = the element's raw Montgomery limbs, most significant limb first (value * 2^256 mod r)",
               source="logs/zolt.log:430-970 ([LASSO PROVER] lines)", rounds=rounds)
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", OUT, len(rounds), "rounds")


if __name__ == "__main__":
    main()
