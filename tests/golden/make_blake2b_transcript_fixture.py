#!/usr/bin/env python3
"""Extracts the Blake2bTranscript preamble the reference printed in its captured run (logs/zolt.log:28-30,1165-1187,
src/transcripts/blake2b.zig:39-180) into tests/golden/blake2b_transcript_preamble.json: the full 32-byte initial state after
init("Jolt") and the sequence of appendU64 / appendBytes(len=0) operations up to the first appendGT (whose 384-byte payload the log
does not hold), with the 8-byte state prefixes the log shows before / after the two empty appendBytes. Data only."""
import json
import os
import re

LOG = "/root/reference/logs/zolt.log"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "blake2b_transcript_preamble.json")


def main():
    lines = open(LOG, errors="replace").read().splitlines()
    i0 = next(i for i, l in enumerate(lines) if "[ZOLT TRANSCRIPT] init:" in l)
    label = re.search(r'label="(\w+)"', lines[i0]).group(1)
    init = bytes(int(x, 16) for x in re.search(r"initial_state=\{ ([0-9a-f ]+)\}", lines[i0 + 2]).group(1).split())
    assert len(init) == 32
    ops = []
    pending = None
    for l in lines[i0 + 3:]:
        if "[ZOLT TRANSCRIPT]" not in l:
            continue
        if "appendGT" in l:
            break
        m = re.search(r"appendU64: value=(\d+)", l)
        if m:
            ops.append({"op": "appendU64", "value": int(m.group(1))})
            continue
        m = re.search(r"appendBytes: len=(\d+), state_before=\{ ([0-9a-f ]+)\.\.\.", l)
        if m:
            assert int(m.group(1)) == 0
            pending = {"op": "appendBytes", "hex": "", "state_before_prefix": m.group(2).replace(" ", "")}
            continue
        m = re.search(r"state_after=\{ ([0-9a-f ]+)\.\.\.", l)
        if m and pending is not None:
            pending["state_after_prefix"] = m.group(1).replace(" ", "")
            ops.append(pending)
            pending = None
    # the state the first appendGT saw = the state after the last operation above
    gt = next(l for l in lines if "appendBytes: len=384" in l)
    final_prefix = re.search(r"state_before=\{ ([0-9a-f ]+)\.\.\.", gt).group(1).replace(" ", "")
    doc = {"source": "reference logs/zolt.log:28-30,1165-1187 (Blake2bTranscript debug prints, src/transcripts/blake2b.zig)",
           "label": label, "initial_state_hex": init.hex(), "ops": ops, "state_prefix_after_all_ops": final_prefix}
    json.dump(doc, open(OUT, "w"), indent=1)
    print("wrote", OUT, len(ops), "ops")


if __name__ == "__main__":
    main()
