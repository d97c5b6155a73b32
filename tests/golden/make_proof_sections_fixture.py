#!/usr/bin/env python3
"""Parses the stage sections of the reference's captured proof file — tests/golden/zolt_proof_regular.bin, written by `zolt prove`
(the standard MultiStageProver path) through serializeProof (src/zkvm/serialization.zig:283-343; the stage records :186-227) — into
tests/golden/proof_stage_sections.json (data only: byte offsets and the parsed field elements, canonical integers as hex).

Layout after the 744-byte header of eleven commitments (serialization.zig:283-306):
  R1CS proof placeholder: tau (u64 count + elements), eval_claims[3], eval_point (count + elements), sumcheck claim, final_eval,
  rounds count, final_point (count + elements);  has_stage_proofs (u8);  log_t, log_k (u64);  then SIX stage records, each
  round_polys (count; per polynomial count + coefficients), challenges (count + elements), final_claims (count + elements).
Field elements are F.toBytes(): the canonical value, 32 little-endian bytes.
The file itself is a committed fixture (data the reference produced); this script only re-reads it."""
import json
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))


def parse(data):
    off = 744

    def u64():
        nonlocal off
        v = struct.unpack_from("<Q", data, off)[0]
        off += 8
        return v

    def fe():
        nonlocal off
        v = int.from_bytes(data[off:off + 32], "little")
        off += 32
        return "%064x" % v

    out = {"source": "tests/golden/zolt_proof_regular.bin (src/zkvm/serialization.zig:283-343)", "header_bytes": 744}
    r1cs_start = off
    tau = [fe() for _ in range(u64())]
    eval_claims = [fe() for _ in range(3)]
    eval_point = [fe() for _ in range(u64())]
    claim, final_eval = fe(), fe()
    n_rounds = u64()
    final_point = [fe() for _ in range(u64())]
    out["r1cs_placeholder"] = {"bytes": [r1cs_start, off], "tau": tau, "eval_claims": eval_claims, "eval_point": eval_point, "claim": claim,
                               "final_eval": final_eval, "rounds": n_rounds, "final_point": final_point}
    out["has_stage_proofs"] = data[off]
    off += 1
    out["log_t"], out["log_k"] = u64(), u64()
    stages = []
    for _ in range(6):
        start = off
        polys = [[fe() for _ in range(u64())] for _ in range(u64())]
        chals = [fe() for _ in range(u64())]
        claims = [fe() for _ in range(u64())]
        stages.append({"bytes": [start, off], "round_polys": polys, "challenges": chals, "final_claims": claims})
    out["stages"] = stages
    assert off == len(data), (off, len(data))
    out["total_bytes"] = off
    # the five commitments the prover absorbs before the first challenge (src/zkvm/mod.zig:421-433), by header offset
    out["absorbed_commitment_offsets"] = {"bytecode": 8, "memory": 232, "memory_final": 296, "registers": 488, "registers_final": 552}
    return out


if __name__ == "__main__":
    with open(os.path.join(HERE, "zolt_proof_regular.bin"), "rb") as fh:
        parsed = parse(fh.read())
    with open(os.path.join(HERE, "proof_stage_sections.json"), "w") as fh:
        json.dump(parsed, fh, indent=0)
    print({k: (v["bytes"], len(v["round_polys"])) for k, v in zip(range(1, 7), parsed["stages"])})
