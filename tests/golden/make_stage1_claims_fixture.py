#!/usr/bin/env python3
"""Extracts the R1CS input claims the reference's captured run appended after Stage 1 — the MLE of each of the first 36 per-cycle R1CS
inputs at r_cycle (R1CSInputEvaluator.computeClaimedInputs, src/zkvm/r1cs/evaluation.zig:55-122) — into
tests/golden/stage1_r1cs_claims.json. Data only.

Source: /root/reference/logs/zolt.log:2092-2277, printed by src/zkvm/proof_converter.zig around the opening-claims block:
  r_cycle[0..8)     [ZOLT MLE] r_cycle[i] (canonical, big-endian); eq_evals[0..3) of the table built from it
  claims[0..36)     every claim as the transcript absorbed it: first / last 8 bytes of its big-endian form, and the full value where the
                    log prints `claim[i] = { ... }` (i = 0..4, 13..20)
  witness samples   witness[0].RightInstructionInput, witness[0].PC, witness[1].LeftInstructionInput, witness[1].PC (little-endian)
Every claim is a linear functional of one column of the 256 x 43 witness matrix: a restatement of R1CSCycleInputs.fromTraceStep
(src/zkvm/r1cs/constraints.zig:930-1223) over the trace regenerated from the ELF is checked column by column against them.

Run in the build container (needs /root/reference); the JSON it writes is committed."""
import json
import os
import re

LOG = "/root/reference/logs/zolt.log"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stage1_r1cs_claims.json")


def braces(line):
    return [bytes(int(x) for x in g.replace(" ", "").strip(",").split(",")).hex() for g in re.findall(r"\{ ?([0-9, ]+?) ?\}", line)]


def main():
    lines = open(LOG, errors="replace").read().splitlines()
    out = {"source": "logs/zolt.log:2092-2277", "r_cycle_be": [], "eq_evals_be": [], "claims": [], "witness_samples_le": {}}
    start = next(i for i, l in enumerate(lines) if "OPENING_CLAIMS: Starting to append 36 claims" in l)
    for l in lines[start - 40:start]:
        m = re.match(r"\[ZOLT MLE\] r_cycle\[(\d+)\] = ", l)
        if m:
            out["r_cycle_be"].append(braces(l)[0])
        m = re.match(r"\[ZOLT MLE\] eq_evals\[(\d+)\] = ", l)
        if m:
            out["eq_evals_be"].append(braces(l)[0])
        m = re.match(r"\[ZOLT\] OPENING_CLAIMS: witness\[(\d+)\]\.(\w+) = ", l)
        if m:
            out["witness_samples_le"]["%s.%s" % m.groups()] = braces(l)[0]
    i = start + 1
    while len(out["claims"]) < 36:
        if lines[i].startswith("[ZOLT TRANSCRIPT] appendBytes: len=32"):
            f8 = re.search(r"first_8_bytes=\{ ([0-9a-f ]+) \}", lines[i + 1]).group(1).replace(" ", "")
            l8 = re.search(r"last_8_bytes=\{ ([0-9a-f ]+) \}", lines[i + 2]).group(1).replace(" ", "")
            rec = {"first8_be": f8, "last8_be": l8}
            m = re.match(r"\[ZOLT\] OPENING_CLAIMS: claim\[(\d+)\] = ", lines[i + 4]) if i + 4 < len(lines) else None
            if m:
                assert int(m.group(1)) == len(out["claims"])
                rec["full_be"] = braces(lines[i + 4])[0]
                assert rec["full_be"][:16] == f8 and rec["full_be"][-16:] == l8
            out["claims"].append(rec)
            i += 3
        else:
            i += 1
    assert len(out["r_cycle_be"]) == 8 and len(out["eq_evals_be"]) == 3 and len(out["witness_samples_le"]) >= 4
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", OUT, sum("full_be" in c for c in out["claims"]), "full claims")


if __name__ == "__main__":
    main()
