#!/usr/bin/env python3
"""Extracts the GruenSplitEqPolynomial data the reference captured in its own run log into
tests/golden/stage4_gruen_eq.json (data only: inputs and expected outputs, no source text).

Source: /root/reference/logs/zolt.log, the [STAGE4_GRUEN_INIT] block (printed by
src/zkvm/spartan/stage4_gruen_prover.zig:266-312):
  r_cycle_be[i]  32 bytes "16 zeros | limbs[2] LE | limbs[3] LE" — the element's raw MONTGOMERY limbs
                 [0, 0, lo, hi] (MontU128Challenge layout, :270-275)
  E_out[0..4]    first entries of the 2^m-entry eq table over w_out = r_cycle_be[0..m], m = n/2
                 (src/poly/split_eq.zig:91-93,122-145), printed with F.toBytes(): canonical value, 32 bytes LE
  E_in[0..4]     the same for w_in = r_cycle_be[m..n-1] (:147-171)
  current_w      toBytes() of w_last = r_cycle_be[n-1]; current_scalar = one

Run in the build container (needs /root/reference); the JSON it writes is committed.
"""
import json
import os
import re
import sys

LOG = "/root/reference/logs/zolt.log"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stage4_gruen_eq.json")


def main():
    lines = open(LOG, errors="replace").read().splitlines()
    start = next(i for i, l in enumerate(lines) if "[STAGE4_GRUEN_INIT] r_cycle_be" in l)
    blk = lines[start:start + 40]
    r_cycle, e_out, e_in = [], [], []
    cur_w = cur_s = None
    n = m = None
    for l in blk:
        mm = re.search(r"r_cycle_be\[(\d+)\] = \{ ([0-9a-f ]+)\}", l)
        if mm:
            b = bytes(int(x, 16) for x in mm.group(2).split())
            assert len(b) == 32 and b[:16] == bytes(16)
            # raw Montgomery limbs [0, 0, lo, hi]
            r_cycle.append([0, 0, int.from_bytes(b[16:24], "little"), int.from_bytes(b[24:32], "little")])
            continue
        mm = re.search(r"n=(\d+), m=(\d+)", l)
        if mm:
            n, m = int(mm.group(1)), int(mm.group(2))
            continue
        mm = re.search(r"(E_out|E_in)\[(\d+)\] = \{ ([0-9, ]+) \}", l)
        if mm:
            b = bytes(int(x) for x in mm.group(3).split(","))
            assert len(b) == 32
            (e_out if mm.group(1) == "E_out" else e_in).append(b.hex())
            continue
        mm = re.search(r"current_(scalar|w \(w_last\)) = \{ ([0-9, ]+) \}", l)
        if mm:
            b = bytes(int(x) for x in mm.group(2).split(","))
            if mm.group(1) == "scalar":
                cur_s = b.hex()
            else:
                cur_w = b.hex()
    assert n == len(r_cycle) == 8 and m == 4 and len(e_out) == 4 and len(e_in) == 4 and cur_w and cur_s
    doc = {
        "source": "reference logs/zolt.log [STAGE4_GRUEN_INIT] block (stage4_gruen_prover.zig:266-312)",
        "n": n, "m": m,
        "r_cycle_be_mont_limbs": [[str(x) for x in row] for row in r_cycle],
        "E_out_len": 16, "E_in_len": 8,
        "E_out_first4_canonical_le_hex": e_out,
        "E_in_first4_canonical_le_hex": e_in,
        "current_scalar_canonical_le_hex": cur_s,
        "current_w_canonical_le_hex": cur_w,
    }
    with open(OUT, "w") as f:
        json.dump(doc, f, indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    sys.exit(main())
