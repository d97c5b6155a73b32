#!/usr/bin/env python3
"""Extracts the Stage-2 BATCHED sumcheck the reference captured in its own run log into tests/golden/stage2_batched_rounds.json
(data only: inputs and expected outputs, no source text).

Source: /root/reference/logs/zolt.log, printed by src/zkvm/batched_sumcheck.zig:127-186 (setupBatching) and :306-420
(generateBatchedProof) with F.toBytes() — the canonical value as 32 little-endian bytes:
  STAGE2_PRE      input_claim[i], num_rounds[i], degree[i] of the five instances, batching_coeff[i] (challengeScalarFull)
  STAGE2_INITIAL  batched_claim = sum_i coeff_i * 2^(max_rounds - rounds_i) * claim_i                     (:161-173)
  STAGE2_ROUND_k  current_claim, the compressed round polynomial c0, c2, c3, the challenge, next_claim     (:334-412)
  STAGE2_FINAL    output_claim
  [ZOLT DEBUG] instK individual_claims[K]: each instance's own claim after its last round (proof_converter.zig, big-endian bytes);
                  the reference checks sum_i coeff_i * individual_i == output_claim ("expected_batched", logs/zolt.log:3583-3585)
A challenge is a MontU128Challenge: its stored Montgomery limbs are [0, 0, lo, hi]; the bytes printed are the canonical value of
that element.

Run in the build container (needs /root/reference); the JSON it writes is committed.
"""
import json
import os
import re

LOG = "/root/reference/logs/zolt.log"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stage2_batched_rounds.json")


def le_bytes(line):
    nums = re.search(r"= \{ ([0-9, ]+)\}", line).group(1)
    b = bytes(int(x) for x in nums.replace(" ", "").strip(",").split(","))
    assert len(b) == 32
    return b.hex()


def main():
    lines = open(LOG, errors="replace").read().splitlines()
    claims, rounds_of, degrees, coeffs = {}, {}, {}, {}
    initial = final = None
    rounds = {}
    individual = {}
    for l in lines:
        m = re.match(r"\[ZOLT\] STAGE2_PRE: input_claim\[(\d+)\] = ", l)
        if m:
            claims[int(m.group(1))] = le_bytes(l)
            continue
        m = re.match(r"\[ZOLT\] STAGE2_PRE: num_rounds\[(\d+)\] = (\d+)", l)
        if m:
            rounds_of[int(m.group(1))] = int(m.group(2))
            continue
        m = re.match(r"\[ZOLT\] STAGE2_PRE: degree\[(\d+)\] = (\d+)", l)
        if m:
            degrees[int(m.group(1))] = int(m.group(2))
            continue
        m = re.match(r"\[ZOLT\] STAGE2_PRE: batching_coeff\[(\d+)\] = ", l)
        if m:
            coeffs[int(m.group(1))] = le_bytes(l)
            continue
        if l.startswith("[ZOLT] STAGE2_INITIAL: batched_claim = "):
            initial = le_bytes(l)
            continue
        if l.startswith("[ZOLT] STAGE2_FINAL: output_claim = "):
            final = le_bytes(l)
            continue
        m = re.match(r"\[ZOLT DEBUG\] inst(\d+) individual_claims\[\d+\] = \{ ([0-9, ]+)\}", l)
        if m:  # printed with toBytesBE (src/zkvm/proof_converter.zig): reversed here to the little-endian form of the rest
            b = bytes(int(x) for x in m.group(2).replace(" ", "").strip(",").split(","))
            assert len(b) == 32
            individual[int(m.group(1))] = b[::-1].hex()
            continue
        m = re.match(r"\[ZOLT\] STAGE2_ROUND_(\d+): (current_claim|c0|c2|c3|challenge|next_claim) = ", l)
        if m:
            rounds.setdefault(int(m.group(1)), {})[m.group(2)] = le_bytes(l)
    # ProductVirtualRemainderProver (instance 0, rounds 16.. of the batch): split_eq.current_scalar before its rounds 0, 1, 2 and the
    # window-table sizes (src/zkvm/spartan/product_remainder.zig:345-356, toBytesBE), and the last Stage-1 r_cycle challenge = the
    # tau the first bind uses (proof_converter.zig OPENING_CLAIMS, toBytes = little-endian)
    product = {"current_scalar": {}, "E_out_len": {}, "E_in_len": {}}
    for l in lines:
        m = re.match(r"\[ZOLT PRODUCT round (\d+)\] split_eq.current_scalar = \{ ([0-9, ]+)\}", l)
        if m:
            b = bytes(int(x) for x in m.group(2).replace(" ", "").strip(",").split(","))
            product["current_scalar"][int(m.group(1))] = b[::-1].hex()
            continue
        m = re.match(r"\[ZOLT PRODUCT round (\d+)\] E_out.len = (\d+), E_in.len = (\d+)", l)
        if m:
            product["E_out_len"][int(m.group(1))] = int(m.group(2))
            product["E_in_len"][int(m.group(1))] = int(m.group(3))
            continue
        m = re.match(r"\[ZOLT\] OPENING_CLAIMS: r_cycle\[last\] = \{ ([0-9, ]+)\}", l)
        if m and "tau_last" not in product:
            product["tau_last"] = bytes(int(x) for x in m.group(1).replace(" ", "").strip(",").split(",")).hex()
        m = re.match(r"\[ZOLT\] OPENING_CLAIMS: r_cycle.len = (\d+)", l)
        if m and "tau_len" not in product:
            product["tau_len"] = int(m.group(1))
    # R1CSInputEvaluator.computeClaimedInputs' debug lines (src/zkvm/r1cs/evaluation.zig:91-103, toBytesBE): the whole Stage-1 r_cycle
    # (= tau_low of the Stage-2 split_eq) and the first three entries of EqPolynomial(r_cycle).evals (256 entries)
    mle = {"r_cycle": {}, "eq_evals": {}}
    for l in lines:
        m = re.match(r"\[ZOLT MLE\] (r_cycle|eq_evals)\[(\d+)\] = \{ ([0-9, ]+)\}", l)
        if m:
            b = bytes(int(x) for x in m.group(3).replace(" ", "").strip(",").split(","))
            assert len(b) == 32
            mle[m.group(1)].setdefault(int(m.group(2)), b[::-1].hex())  # first occurrence; reversed to little-endian
    # computeOpeningClaims' eq table (src/zkvm/spartan/product_remainder.zig:396-425, computeEqEvalsGeneric :496-531; printed big-endian):
    # r_cycle there = the last n_cycle_vars batch challenges reversed; first three entries of its 256-entry table
    factor_eq = {}
    for l in lines:
        m = re.match(r"\[ZOLT\] FACTOR_EVALS: eq_evals\[(\d+)\] = \{ ([0-9, ]+)\}", l)
        if m:
            b = bytes(int(x) for x in m.group(2).replace(" ", "").strip(",").split(","))
            factor_eq.setdefault(int(m.group(1)), b[::-1].hex())
    # OutputSumcheckProver (instance 3, rounds 8.. of the batch; src/zkvm/ram/output_check.zig:100-365 init, :586-609 eq table, printed
    # big-endian): its r_address, the region bounds it derived, and the five folded tables' final values after its 16 rounds
    oc = {"r_address": {}, "final": {}}
    for l in lines:
        m = re.match(r"\[ZOLT OUTPUT_CHECK\]\s+r_address\[(\d+)\] = \{ ([0-9, ]+)\}", l)
        if m:
            oc["r_address"].setdefault(int(m.group(1)), bytes(int(x) for x in m.group(2).replace(" ", "").strip(",").split(","))[::-1].hex())
            continue
        m = re.match(r"\[ZOLT OUTPUT_CHECK\] (val_final|val_init|val_io|eq_r_address|io_mask)\[0\]: \{ ([0-9, ]+)\}", l)
        if m:
            oc["final"].setdefault(m.group(1), bytes(int(x) for x in m.group(2).replace(" ", "").strip(",").split(","))[::-1].hex())
            continue
        m = re.match(r"\[ZOLT\] OutputSumcheck: lowest=0x[0-9A-Fa-f]+, io_start=(\d+), io_end=(\d+)", l)
        if m:
            oc["io_start"], oc["io_end"] = int(m.group(1)), int(m.group(2))
            continue
        m = re.match(r"\[ZOLT\] OutputSumcheck: termination_index=(\d+)", l)
        if m:
            oc["termination_index"] = int(m.group(1))
            continue
        m = re.match(r"\[ZOLT\] OutputSumcheck: final_ram non_zero_count=(\d+), io_region_values=(\d+), K=(\d+)", l)
        if m:
            oc["ram_words"], oc["K"] = int(m.group(1)), int(m.group(3))
            continue
        m = re.match(r"\[ZOLT\] OutputSumcheck: initial_ram k=(\d+), addr=0x80000000,", l)
        if m:
            oc["first_ram_index"] = int(m.group(1))
    assert sorted(oc["r_address"]) == list(range(16)) and len(oc["final"]) == 5 and oc["K"] == 1 << 16
    n = len(claims)
    assert n == 5 and sorted(rounds) == list(range(max(rounds_of.values())))
    assert sorted(product["current_scalar"]) == [0, 1, 2] and sorted(factor_eq) == [0, 1, 2]
    assert sorted(mle["r_cycle"]) == list(range(product["tau_len"])) and sorted(mle["eq_evals"]) == [0, 1, 2]
    assert mle["r_cycle"][product["tau_len"] - 1] == product["tau_last"]
    gl = next(l for l in lines if l.startswith("[ZOLT] STAGE2_BATCHED: gamma_instr = "))
    gamma_instr = bytes(int(x) for x in re.search(r"= \{ ([0-9, ]+)\}", gl).group(1).replace(" ", "").strip(",").split(",")).hex()
    out = {
        "source": "reference logs/zolt.log, STAGE2_* lines of src/zkvm/batched_sumcheck.zig (canonical little-endian hex)",
        "input_claims": [claims[i] for i in range(n)],
        "num_rounds": [rounds_of[i] for i in range(n)],
        "degrees": [degrees[i] for i in range(n)],
        "batching_coeffs": [coeffs[i] for i in range(n)],
        "initial_batched_claim": initial,
        "rounds": [rounds[k] for k in sorted(rounds)],
        "output_claim": final,
        "instance_final_claims": [individual[i] for i in range(n)],
        "gamma_instr_be": gamma_instr,  # [ZOLT] STAGE2_BATCHED: gamma_instr (canonical, big-endian), proof_converter.zig:2790
        "product_remainder": {
            "tau_len": product["tau_len"],
            "tau_last": product["tau_last"],
            "current_scalar_before_round": [product["current_scalar"][k] for k in range(3)],
            "E_out_len": [product["E_out_len"][k] for k in range(3)],
            "E_in_len": [product["E_in_len"][k] for k in range(3)],
            "first_batch_round": max(rounds_of.values()) - rounds_of[0],
        },
        "stage1_r_cycle": [mle["r_cycle"][i] for i in range(product["tau_len"])],
        "eq_evals_of_r_cycle_first3": [mle["eq_evals"][i] for i in range(3)],
        "eq_evals_of_reversed_stage2_challenges_first3": [factor_eq[i] for i in range(3)],
        "output_check": {
            "K": oc["K"],
            "io_start": oc["io_start"],
            "io_end": oc["io_end"],
            "termination_index": oc["termination_index"],
            "first_ram_index": oc["first_ram_index"],
            "ram_words": oc["ram_words"],
            "r_address": [oc["r_address"][i] for i in range(16)],
            "final": oc["final"],
            "first_batch_round": max(rounds_of.values()) - rounds_of[3],
        },
    }
    for r in out["rounds"]:
        assert set(r) == {"current_claim", "c0", "c2", "c3", "challenge", "next_claim"}
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", OUT, len(out["rounds"]), "rounds")


if __name__ == "__main__":
    main()
