#!/usr/bin/env python3
"""Extracts the Stage-3 batched sumcheck (ShiftSumcheck + InstructionInput + RegistersClaimReduction, 8 rounds) the reference captured
in its own run log into tests/golden/stage3_batched_rounds.json (data only: inputs and expected outputs, no source text).

Source: /root/reference/logs/zolt.log, printed by src/zkvm/spartan/stage3_prover.zig:113-760 with F.toBytes() — the canonical value
as 32 little-endian bytes:
  challengeScalarFull after "STAGE 3 BEGIN": the Shift gamma, the InstructionInput gamma, the Registers gamma, then (after the three
                  input claims were appended) the three batching coefficients — canonical 128-bit values
  STAGE3_PRE      input_claim[0..2]
  STAGE3_ROUND_k  shift_p0, shift_p1 (ShiftSumcheck's own evaluations), the combined c0, c2, c3, the challenge, next_claim
  STAGE3_DEBUG / STAGE3_OPENING / STAGE3->4: final claims of the three instances and the final combined claim
  STAGE3: r_product[0] (toBytesBE: the 8 most significant bytes), SHIFT_INIT: r_outer[0], r_outer[last] (the 8 least significant bytes) of entries of the two opening points (a check on how
                  they are derived from the Stage-1 and Stage-2 fixtures)
Run in the build container (needs /root/reference); the JSON it writes is committed."""
import json
import os
import re

LOG = "/root/reference/logs/zolt.log"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stage3_batched_rounds.json")


def nums(line):
    body = re.search(r"\{ \{? ?([0-9, ]+?) ?\}? \}", line).group(1)
    return bytes(int(x) for x in body.replace(" ", "").strip(",").split(","))


def main():
    lines = open(LOG, errors="replace").read().splitlines()
    begin = next(i for i, l in enumerate(lines) if "STAGE 3 BEGIN" in l)
    end = next(i for i, l in enumerate(lines) if l.startswith("[ZOLT STAGE3->4]   rs2_value"))
    full = []
    for i in range(begin, end):
        if "challengeScalarFull" in lines[i]:
            full.append(re.search(r"canonical_value=0x([0-9a-f]+)", lines[i + 3]).group(1))
        if lines[i].startswith("[ZOLT] STAGE3_ROUND_0"):
            break
    assert len(full) == 6, full
    out = {"source": "logs/zolt.log (stage3_prover.zig:113-760)", "shift_gamma_be": full[0], "instr_gamma_be": full[1], "reg_gamma_be": full[2],
           "batching_coeffs_be": full[3:], "input_claims": [None] * 3, "rounds": {}, "final": {}, "prefix8": {}}
    for l in lines[begin - 10:end + 1]:
        m = re.match(r"\[ZOLT\] STAGE3_PRE: input_claim\[(\d)\]", l)
        if m:
            out["input_claims"][int(m.group(1))] = nums(l).hex()
            continue
        m = re.match(r"\[ZOLT\] STAGE3_ROUND_(\d+): (shift_p0|shift_p1|c0|c2|c3|challenge|next_claim) = ", l)
        if m:
            b = nums(l)
            assert len(b) == 32
            out["rounds"].setdefault(int(m.group(1)), {})[m.group(2)] = b.hex()
            continue
        m = re.match(r"\[ZOLT\] STAGE3_OPENING: (unexpanded_pc|pc|is_noop) = ", l)
        if m:
            out["final"]["shift_" + m.group(1)] = nums(l).hex()
            continue
        m = re.match(r"\[ZOLT STAGE3->4\]   (rd_write_value|rs1_value|rs2_value) = \{ ([0-9, ]+)\}", l)
        if m:
            out["final"]["reg_" + m.group(1)] = bytes(int(x) for x in m.group(2).replace(" ", "").strip(",").split(",")).hex()
            continue
        m = re.match(r"\[ZOLT\] STAGE3_DEBUG: (current_instr_claim|current_reg_claim|combined_claim) = ", l)
        if m:
            out["final"][m.group(1)] = nums(l).hex()
            continue
        m = re.match(r"\[ZOLT\] STAGE3: (r_product)\[0\] = \{ ([0-9, ]+)\}", l)
        if m:
            out["prefix8"]["r_product_0"] = bytes(int(x) for x in m.group(2).replace(" ", "").strip(",").split(",")).hex()
            continue
        m = re.match(r"\[ZOLT\] SHIFT_INIT: r_outer\[(0|last)\] = ", l)
        if m:
            out["prefix8"]["r_outer_" + m.group(1)] = nums(l).hex()
    out["rounds"] = [out["rounds"][k] for k in sorted(out["rounds"])]
    assert len(out["rounds"]) == 8 and all(len(r) == 7 for r in out["rounds"]), [len(r) for r in out["rounds"]]
    assert all(out["input_claims"]) and len(out["final"]) == 9 and len(out["prefix8"]) == 3, (out["final"].keys(), out["prefix8"])
    json.dump(out, open(OUT, "w"), indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
