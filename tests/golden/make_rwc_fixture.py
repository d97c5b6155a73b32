#!/usr/bin/env python3
"""Extracts what the reference's captured run printed about its RamReadWriteCheckingProver (Stage-2 instance 2: log_k = 16 address +
log_t = 8 cycle variables, three phases 4 / 16 / 4) into tests/golden/rwc_captured_run.json — data only, no source text.

Source: /root/reference/logs/zolt.log:2397,2635-3700, printed by src/zkvm/ram/read_write_checking.zig (init :189-376, the phase
polynomials :410-769, the binds :902-1185, getOpeningClaims :1210-1322) and src/zkvm/proof_converter.zig (gamma_rwc :2777, the three
opening claims). Everything the prover consumed is known:
  gamma (full), r_cycle = the eight Stage-1 challenges (tests/golden/stage2_batched_rounds.json: stage1_r_cycle; the log's tau[0] /
  tau[last] prefixes identify the order), K = 2^16, start address, phase1_num_rounds = log_t / 2, the initial RAM (the ELF's 13 code
  words at index 4096..4108: tests/golden/fibonacci.elf), the one memory access of the run (a write of 1 at cycle 54, index 2049 — the
  termination bit), input claim 0, and the 24 Stage-2 challenges (same fixture). Printed outputs: per cycle-phase round the 8-byte
  big-endian prefixes of q_constant, q_quadratic, the claim, s(0), s(1); the bound entry after every bind; the first address rounds'
  s(0), s(1), s(2); eq_cycle_scalar / inc_scalar at the phase switch; the reordered challenges of getOpeningClaims (8-byte
  little-endian prefixes); and ra_claim, val_claim, inc_claim in full.

Run in the build container (needs /root/reference); the JSON it writes is committed."""
import json
import os
import re

LOG = "/root/reference/logs/zolt.log"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rwc_captured_run.json")


def ints(text):
    return [int(x) for x in re.findall(r"\d+", text)]


def braces(line):
    return [bytes(ints(g)).hex() for g in re.findall(r"\{([0-9, ]+)\}", line)]


def main():
    lines = open(LOG, errors="replace").read().splitlines()
    out = {"source": "logs/zolt.log:2397,2635-3700", "rounds": {}, "init": {"val_init_shown": [], "entries": []}, "opening": {"r_cycle_le8": [], "r_address_le8": []}}

    def rd(k):
        return out["rounds"].setdefault(str(k), {})
    cur = None
    for l in lines:
        if l.startswith("[ZOLT] STAGE2_BATCHED: gamma_rwc = "):
            out["gamma_be"] = braces(l)[0]
        m = re.match(r"\[RWC INIT\] params.start_address = 0x([0-9a-f]+)", l)
        if m:
            out["start_address"] = int(m.group(1), 16)
        m = re.match(r"\[RWC INIT\] K = (\d+), initial_ram entries = (\d+)", l)
        if m:
            out["K"], out["initial_ram_entries"] = int(m.group(1)), int(m.group(2))
        m = re.match(r"\[RWC INIT\]   addr=0x([0-9a-f]+), idx=(\d+), val=(\d+)", l)
        if m:
            out["init"]["val_init_shown"].append({"addr": int(m.group(1), 16), "idx": int(m.group(2)), "val": int(m.group(3))})
        m = re.match(r"\[RWC INC SET\] cycle=(\d+), new_val=(\d+), prev_val=(\d+)", l)
        if m:
            out["init"]["inc_set"] = {"cycle": int(m.group(1)), "new_val": int(m.group(2)), "prev_val": int(m.group(3)), "inc_be": braces(l)[0]}
        m = re.match(r"\[RWC INIT\] entry: cycle=(\d+), addr=(\d+), op=(\d+), prev_val=(\d+), next_val=(\d+)", l)
        if m:
            out["init"]["entries"].append(dict(zip(("cycle", "addr", "op", "prev_val", "next_val"), map(int, m.groups()))))
        m = re.match(r"\[RWC INIT\] tau\[(0|last)\] = ", l)
        if m:
            out["init"]["tau_" + m.group(1) + "_be8"] = braces(l)[0]
        m = re.match(r"\[RWC PHASE1\] round=(\d+), q_constant=", l)
        if m:
            cur = int(m.group(1))
            rd(cur)["phase"] = "cycle"
            rd(cur)["q_constant_be8"] = braces(l)[0]
        if l.startswith("[RWC PHASE1] q_quadratic=") and cur is not None:
            rd(cur)["q_quadratic_be8"], rd(cur)["claim_be8"] = braces(l)
        if l.startswith("[RWC PHASE1] result: s0=") and cur is not None:
            rd(cur)["s0_be8"], rd(cur)["s1_be8"] = braces(l)
        m = re.match(r"\[RWC BIND\] round=(\d+), entries.len after bind=(\d+)", l)
        if m:
            cur = int(m.group(1))
            rd(cur)["entries_after_bind"] = int(m.group(2))
        m = re.match(r"\[RWC BIND\]   entry\[0\]: cycle=(\d+), addr=(\d+), ra_coeff=", l)
        if m and cur is not None:
            rd(cur)["entry0_after_bind"] = {"cycle": int(m.group(1)), "addr": int(m.group(2)), "ra_coeff_be8": braces(l)[0]}
        m = re.match(r"\[RWC PHASE2\] round=(\d+), addr_round=(\d+), entries.len=(\d+)", l)
        if m:
            cur = int(m.group(1))
            rd(cur).update(phase="address", addr_round=int(m.group(2)), entries=int(m.group(3)))
        if l.startswith("[RWC PHASE2] eq_cycle_scalar = "):
            out["phase2_eq_cycle_scalar_be8"] = braces(l)[0]
        if l.startswith("[RWC PHASE2] inc_scalar = "):
            out["phase2_inc_scalar_be8"] = braces(l)[0]
        m = re.match(r"\[RWC PHASE2\] entry\[0\]: addr=(\d+), ra_coeff=", l)
        if m:
            b = braces(l)
            out["phase2_entry0"] = {"addr": int(m.group(1)), "ra_coeff_be8": b[0], "val_coeff_be8": b[1]}
        if l.startswith("[RWC PHASE2] result: s0=") and cur is not None:
            rd(cur)["s0_be8"], rd(cur)["s1_be8"], rd(cur)["s2_be8"] = braces(l)
        m = re.match(r"\[RWC BIND PHASE2\] addr_round=(\d+), entries.len after bind=(\d+)", l)
        if m and cur is not None:
            rd(cur)["entries_after_bind"] = int(m.group(2))
        m = re.match(r"\[RWC GET_OPENING\]   r_cycle\[(\d+)\] = ", l)
        if m:
            out["opening"]["r_cycle_le8"].append(braces(l)[0])
        m = re.match(r"\[RWC GET_OPENING\]   r_address\[(\d+)\] = ", l)
        if m:
            out["opening"]["r_address_le8"].append(braces(l)[0])
        m = re.match(r"\[RWC GET_OPENING\] phase1_end=(\d+), phase2_end=(\d+), phase3_cycle_len=(\d+)", l)
        if m:
            out["phase1_num_rounds"], out["phase2_end"] = int(m.group(1)), int(m.group(2))
        if l.startswith("[RWC GET_OPENING] val_claim base (bound val_init[0]) = "):
            out["opening"]["val_claim_base_le8"] = braces(l)[0]
        if l.startswith("[RWC GET_OPENING] inc_claim (bound inc[0]) = "):
            out["opening"]["inc_claim_le8"] = braces(l)[0]
        for k in ("ra_claim", "val_claim", "inc_claim"):
            if l.startswith(f"[ZOLT] STAGE2 RWC: {k} = "):
                out["opening"][k + "_be"] = braces(l)[0]
    out["log_k"], out["log_t"] = 16, 8
    assert len(out["rounds"]) == 24 and len(out["opening"]["r_cycle_le8"]) == 8 and len(out["opening"]["r_address_le8"]) == 16
    assert out["init"]["entries"] == [{"cycle": 54, "addr": 2049, "op": 1, "prev_val": 0, "next_val": 1}]
    out["encoding"] = "*_be: big-endian bytes of the canonical value (toBytesBE); *_be8: their first 8; *_le8: the first 8 little-endian bytes (toBytes)"
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
