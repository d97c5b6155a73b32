#!/usr/bin/env python3
"""Extracts the Stage-2 UniSkip first round (product virtualisation, five product constraints over the domain {-2..2}) the reference's
captured run printed into tests/golden/stage2_uniskip.json. Data only.

Source: /root/reference/logs/zolt.log:2276-2330, printed by src/zkvm/proof_converter.zig:1083-1170 and :4257-4305:
  tau_high           the freshly sampled challenge (big-endian);  tau = [Stage-1 r_cycle reversed, tau_high] (:1112-1137)
  base_evals[0..5)   the five product claims of Stage 1 (Product, WriteLookupOutputToRD, WritePCtoRD, ShouldBranch, ShouldJump; big-endian)
  extended_evals[4]  t1 at the targets -3, 3, -4, 4 (big-endian)
  coeffs[0..13)      s1(Y) = L(tau_high, Y) t1(Y) (printed little-endian)
  input_claim, r0, uni_skip_claim = s1(r0) (big-endian)
Run in the build container (needs /root/reference); the JSON it writes is committed."""
import json
import os
import re

LOG = "/root/reference/logs/zolt.log"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stage2_uniskip.json")


def braces(line):
    return [bytes(int(x) for x in g.replace(" ", "").strip(",").split(",")).hex() for g in re.findall(r"\{ ?([0-9, ]+?) ?\}", line)]


def main():
    out = {"source": "logs/zolt.log:2276-2330", "base_evals_be": [], "extended_evals_be": [], "coeffs_le": []}
    for l in open(LOG, errors="replace").read().splitlines():
        for key, pat in (("tau_high_be", r"\[ZOLT\] STAGE2: tau_high = "), ("input_claim_be", r"\[ZOLT\] STAGE2_UNISKIP: input_claim = "),
                         ("r0_be", r"\[ZOLT\] STAGE2: r0 = "), ("uni_skip_claim_be", r"\[ZOLT\] STAGE2: uni_skip_claim = ")):
            if re.match(pat, l) and key not in out:
                out[key] = braces(l)[0]
        for key, pat in (("base_evals_be", r"\[ZOLT\] STAGE2: base_evals\[(\d+)\] = "), ("extended_evals_be", r"\[ZOLT\] STAGE2_UNISKIP: extended_evals\[(\d+)\] = "),
                         ("coeffs_le", r"\[ZOLT\] STAGE2_UNISKIP: coeffs\[(\d+)\] = ")):
            m = re.match(pat, l)
            if m and int(m.group(1)) == len(out[key]):
                out[key].append(braces(l)[0])
    assert len(out["base_evals_be"]) == 5 and len(out["extended_evals_be"]) == 4 and len(out["coeffs_le"]) == 13
    assert all(k in out for k in ("tau_high_be", "input_claim_be", "r0_be", "uni_skip_claim_be"))
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
