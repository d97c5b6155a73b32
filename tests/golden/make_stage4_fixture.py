#!/usr/bin/env python3
"""Extracts what the reference's captured run printed about its Stage4GruenProver (RegistersReadWriteChecking: K = 128 registers,
T = 256 cycles, phases 4 / 7 / 4) into tests/golden/stage4_registers_run.json — data only, no source text.

Source: /root/reference/logs/zolt.log:4416-5290, printed by src/zkvm/spartan/stage4_gruen_prover.zig and the Stage-4 loop of
src/zkvm/proof_converter.zig:
  gamma                   [STAGE4] gamma_full_BE (canonical value, big-endian)
  r_cycle_be[0..8)        with tests/golden/stage4_gruen_eq.json (raw limbs [0, 0, lo, hi]); round order = reversed
  phase config            4 cycle / 7 address / 4 cycle rounds
  input claim             [PROOF_CONV STAGE4] regs_current_claim before round 0 (toBytes: canonical, little-endian), and the
                          registers instance's own evaluations p(0..3) of round 0 in full
  challenges              the 15 batched-sumcheck challenges, full (little-endian)
  final                   merged_eq[0] = eq_scalar, combined = ra*val + wa*(val + inc), expected = their product, and the instance's
                          final claim (little-endian, full)
The trace itself is regenerated from tests/golden/fibonacci.elf by the interpreter in tests/util.py (54 executed steps, 202 no-ops).

Run in the build container (needs /root/reference); the JSON it writes is committed."""
import json
import os
import re

LOG = "/root/reference/logs/zolt.log"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stage4_registers_run.json")


def braces(line):
    return [bytes(int(x) for x in g.replace(" ", "").strip(",").split(",")).hex() for g in re.findall(r"\{ ?([0-9, ]+?) ?\}", line)]


def main():
    lines = open(LOG, errors="replace").read().splitlines()
    out = {"source": "logs/zolt.log:4416-5290", "challenges_le": {}, "round0": {}, "final": {}}
    for i, l in enumerate(lines):
        if l.startswith("[STAGE4] gamma_full_BE = "):
            out["gamma_be"] = braces(l)[0]
        m = re.match(r"\[STAGE4 INIT\] Phase config: phase1=(\d+), phase2=(\d+), phase3_cycle=(\d+)", l)
        if m:
            out["phase1_num_rounds"], out["phase2_num_rounds"], out["phase3_cycle_rounds"] = map(int, m.groups())
        m = re.match(r"\[STAGE4 INIT\] T=(\d+), K=(\d+)", l)
        if m:
            out["T"], out["K"] = int(m.group(1)), int(m.group(2))
        if l.startswith("[STAGE4 INIT] inc_poly first 4:"):
            out["inc_first4_le8"] = braces(l)
        if l.startswith("[PROOF_CONV STAGE4]   regs_current_claim = ") and "claim_le" not in out["round0"]:
            out["round0"]["claim_le"] = braces(l)[0]
        m = re.match(r"\[PROOF_CONV STAGE4\]   regs_evals\[(\d)\] = ", l)
        if m and len(out["round0"]) < 5:
            out["round0"]["p%s_le" % m.group(1)] = braces(l)[0]
        m = re.match(r"\[ZOLT STAGE4\] Round (\d+): challenge \(LE\) = ", l)
        if m:
            out["challenges_le"][m.group(1)] = braces(l)[0]
        if l.startswith("[ZOLT STAGE4 FINAL BIND] eq_scalar = "):
            out["final"]["eq_scalar_le"] = braces(l)[0]
        if l.startswith("[ZOLT STAGE4 FINAL BIND] combined (ra*val + wa*(val+inc)) = "):
            out["final"]["combined_le"] = braces(l)[0]
        if l.startswith("[ZOLT STAGE4 FINAL BIND] expected (eq * combined) = "):
            out["final"]["expected_le"] = braces(l)[0]
        if l.startswith("[ZOLT STAGE4 FINAL DEBUG] regs_current_claim (poly_0 final) = "):
            out["final"]["claim_le"] = braces(l)[0]
    assert len(out["challenges_le"]) == 15 and len(out["round0"]) == 5 and len(out["final"]) == 4
    out["challenges_le"] = [out["challenges_le"][str(k)] for k in range(15)]
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
