"""Shared helpers for the tests: representation conversions and seeded inputs.

Conversions go through oracle/pymodel.py big-ints (test infrastructure)."""
import json
import os

import numpy as np

from oracle import pymodel as pm

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
M64 = (1 << 64) - 1


def load_vectors():
    with open(os.path.join(GOLDEN, "vectors.json")) as f:
        return json.load(f)


def mont_limbs(v, mod):
    """canonical int -> (4,) uint64 Montgomery limbs"""
    return np.array(pm.limbs(pm.to_mont(v, mod)), dtype=np.uint64)


def fr(vals):
    return np.array([pm.limbs(pm.to_mont(int(v), pm.R_MOD)) for v in vals], dtype=np.uint64).reshape(-1, 4)


def fp(vals):
    return np.array([pm.limbs(pm.to_mont(int(v), pm.P_MOD)) for v in vals], dtype=np.uint64).reshape(-1, 4)


def fr_hex(hexes):
    return fr([int(h, 16) for h in hexes])


def fr_to_int(l):
    return pm.from_mont(pm.from_limbs(l), pm.R_MOD)


def fp_to_int(l):
    return pm.from_mont(pm.from_limbs(l), pm.P_MOD)


def points_xy(pts):
    """list of (x,y) canonical ints -> (n,8) uint64 Montgomery"""
    out = np.zeros((len(pts), 8), dtype=np.uint64)
    for i, p in enumerate(pts):
        if p is None:
            continue
        out[i, :4] = pm.limbs(pm.to_mont(p[0], pm.P_MOD))
        out[i, 4:] = pm.limbs(pm.to_mont(p[1], pm.P_MOD))
    return out


def point_from_xy(xy, inf):
    if inf:
        return None
    return (fp_to_int(xy[:4]), fp_to_int(xy[4:]))


def splitmix64(seed, n):
    """n uint64 words of splitmix64(seed) (SURVEY §8(d) synthetic inputs) — vectorised."""
    idx = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def random_raw256(seed, n):
    """(n,4) uint64 raw 256-bit integers (may exceed the modulus)."""
    return splitmix64(seed, 4 * n).reshape(n, 4)


def fibonacci_rd_values(elf_bytes, steps=54, records=None):
    """rd_value of every step of the reference's captured fibonacci run (logs/zolt.log:23-27: 54 cycles, terminated by the
    `j .` at 0x80000010), from a minimal RV64 interpreter of the ten instruction forms the 104-byte program uses — the semantics
    of the reference's tracer (src/tracer/mod.zig:429-816: rd_value = the value computed for rd, 0 for branches; JAL / JALR
    record pc + 4 even when rd = x0). Test infrastructure for the register-commitment fixture (src/zkvm/mod.zig:1585-1617)."""
    import struct
    code = elf_bytes[0x1000:0x1000 + 104]
    mask = (1 << 64) - 1

    def sx(v, b):
        return v - (1 << b) if v >> (b - 1) else v

    regs, pc, vals = [0] * 32, 0x80000000, []
    for _ in range(steps):
        w = struct.unpack("<I", code[pc - 0x80000000:pc - 0x80000000 + 4])[0]
        op, rd, f3, rs1, rs2, f7 = w & 0x7F, (w >> 7) & 31, (w >> 12) & 7, (w >> 15) & 31, (w >> 20) & 31, w >> 25
        imm_i, npc, rv = sx(w >> 20, 12), pc + 4, 0
        a, b = regs[rs1], regs[rs2]
        if op == 0x37:  # LUI
            rv = sx(w & 0xFFFFF000, 32) & mask
        elif op == 0x1B and f3 == 0:  # ADDIW
            rv = sx((a + imm_i) & 0xFFFFFFFF, 32) & mask
        elif op == 0x13 and f3 == 0:  # ADDI
            rv = (a + imm_i) & mask
        elif op == 0x13 and f3 == 1:  # SLLI
            rv = (a << ((w >> 20) & 0x3F)) & mask
        elif op == 0x6F:  # JAL
            imm = sx(((w >> 31) << 20) | (((w >> 12) & 0xFF) << 12) | (((w >> 20) & 1) << 11) | (((w >> 21) & 0x3FF) << 1), 21)
            rv, npc = pc + 4, pc + imm
        elif op == 0x67:  # JALR
            rv, npc = pc + 4, (a + imm_i) & ~1
        elif op == 0x63:  # BRANCH: no rd, rd_value stays 0
            imm = sx(((w >> 31) << 12) | (((w >> 7) & 1) << 11) | (((w >> 25) & 0x3F) << 5) | (((w >> 8) & 0xF) << 1), 13)
            sa, sb = sx(a, 64), sx(b, 64)
            if {0: a == b, 1: a != b, 4: sa < sb, 5: sa >= sb, 6: a < b, 7: a >= b}[f3]:
                npc = pc + imm
            rd = 0
        elif op == 0x3B and f3 == 0:  # ADDW / SUBW
            rv = sx(((a - b) if f7 & 0x20 else (a + b)) & 0xFFFFFFFF, 32) & mask
        else:
            raise ValueError(f"unexpected instruction {w:08x} at {pc:x}")
        if rd:
            regs[rd] = rv
        vals.append(rv)
        if records is not None:
            records.append((w, rv, a, b, pc))
        pc = npc
    assert pc == 0x80000010 and regs[10] == 55  # back in the `j .` loop with fib = 55 in a0
    return vals


def fibonacci_trace_steps(elf_bytes, steps=54, padded=256):
    """ExecutionTrace.steps of the captured run as Stage4GruenProver reads them (src/zkvm/spartan/stage4_gruen_prover.zig:183-246):
    (instruction word, rd_value, is_noop) per cycle — the 54 executed instructions, then no-op padding up to the trace length."""
    recs = []
    fibonacci_rd_values(elf_bytes, steps, recs)
    return [(w, rv, False) for w, rv, _, _, _ in recs] + [(0, 0, True)] * (padded - len(recs))


def fibonacci_full_trace(elf_bytes, steps=54, padded=256):
    """the same run as tracer.TraceStep records (src/tracer/mod.zig:14-45, 300-318): what R1CSCycleInputs.fromTraceStep reads of a step —
    instruction, pc (= unexpanded_pc: no virtual sequences), rs1_value / rs2_value (the registers named by the instruction's fields, read
    before it executes), rd_value, no memory access, not compressed — then NoOp padding (padWithNoop)."""
    recs = []
    fibonacci_rd_values(elf_bytes, steps, recs)
    out = [{"instruction": w, "pc": pc, "unexpanded_pc": pc, "rs1_value": a, "rs2_value": b, "rd_value": rv, "memory_value": None, "is_compressed": False,
            "is_noop": False} for w, rv, a, b, pc in recs]
    noop = {"instruction": 0, "pc": 0, "unexpanded_pc": 0, "rs1_value": 0, "rs2_value": 0, "rd_value": 0, "memory_value": None, "is_compressed": False, "is_noop": True}
    return out + [dict(noop) for _ in range(padded - len(out))]


def fibonacci_lookup_indices(elf_bytes, steps=54):
    """LookupTraceCollector.recordInstruction's entries for the captured fibonacci run (src/zkvm/instruction/lookup_trace.zig:843-1015, the
    tracer records one per instruction BEFORE it executes, src/tracer/mod.zig:269-277), as the u128 lookup indices LassoProver reads
    (src/zkvm/prover.zig:609-612) — 44 entries for the 54 cycles: ADDIW (opcode OP_IMM_32) is not recorded. Per form
    (src/zkvm/instruction/lookups.zig): ADDI -> the wrapped sum rs1 + imm (AddLookup.toLookupIndex :49-54); SLLI -> interleaveBits(rs1,
    shamt) (:982); BNE -> interleaveBits(rs1, rs2) (:406-408); ADDW -> the sign-extended 32-bit sum (:1706), SUBW -> the interleaved low
    words (:1757); JAL / JALR -> pc + 4 (:720, :777); LUI -> the sign-extended immediate (:622-630). interleaveBits (lookup_table/mod.zig:52-73)
    puts x on the odd and y on the even bit positions. Test infrastructure for the proof-file fixture: -> (n, 2) uint64 (low, high)."""
    import numpy as np
    recs = []
    fibonacci_rd_values(elf_bytes, steps, recs)
    mask = (1 << 64) - 1

    def sx(v, b):
        return v - (1 << b) if v >> (b - 1) else v

    def interleave(x, y):
        r = 0
        for i in range(64):
            r |= ((x >> i) & 1) << (2 * i + 1)
            r |= ((y >> i) & 1) << (2 * i)
        return r

    idx = []
    for w, _, a, b, pc in recs:
        op, f3, f7, imm_i = w & 0x7F, (w >> 12) & 7, w >> 25, sx(w >> 20, 12)
        if op == 0x13 and f3 == 0:
            idx.append((a + imm_i) & mask)
        elif op == 0x13 and f3 == 1:
            idx.append(interleave(a, (w >> 20) & 0x3F))
        elif op == 0x3B and f3 == 0:
            idx.append(interleave(a & 0xFFFFFFFF, b & 0xFFFFFFFF) if f7 & 0x20 else sx((a + b) & 0xFFFFFFFF, 32) & mask)
        elif op == 0x63:
            idx.append(interleave(a, b))
        elif op in (0x6F, 0x67):
            idx.append((pc + 4) & mask)
        elif op == 0x37:
            idx.append(sx(w & 0xFFFFF000, 32) & mask)
        elif op != 0x1B:
            raise ValueError(f"unexpected instruction {w:08x}")
    return np.array([[v & mask, v >> 64] for v in idx], dtype=np.uint64)


def proof_file_sections():
    """tests/golden/proof_stage_sections.json (make_proof_sections_fixture.py): the stage records of the reference's captured proof file"""
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "proof_stage_sections.json")) as fh:
        return json.load(fh)


def serialize_stage_sections(log_t, log_k, stages):
    """the bytes serializeProof writes after the 744-byte header (src/zkvm/serialization.zig:308-343, stage records :186-227) for the
    R1CS placeholder the standard prover emits (R1CSProof.placeholder: one zero tau, three zero claims, one zero eval point, zero claim /
    final_eval, no rounds, empty final point) and six stage records; stages: [(round_polys, challenges, final_claims)] of canonical ints"""
    import struct

    def fe(v):
        return int(v).to_bytes(32, "little")

    out = bytearray()
    out += struct.pack("<Q", 1) + fe(0) + fe(0) * 3 + struct.pack("<Q", 1) + fe(0) + fe(0) + fe(0) + struct.pack("<Q", 0) + struct.pack("<Q", 0)
    out += b"\x01" + struct.pack("<QQ", log_t, log_k)
    for polys, chals, claims in stages:
        out += struct.pack("<Q", len(polys))
        for p in polys:
            out += struct.pack("<Q", len(p)) + b"".join(fe(c) for c in p)
        out += struct.pack("<Q", len(chals)) + b"".join(fe(c) for c in chals)
        out += struct.pack("<Q", len(claims)) + b"".join(fe(c) for c in claims)
    return bytes(out)


def output_check_tables_of_the_captured_run(oc, elf_bytes, fr_from_int, eq_table):
    """The five tables OutputSumcheckProver.init (src/zkvm/ram/output_check.zig:100-365) built in the reference's captured fibonacci
    run, from what logs/zolt.log states about it (tests/golden/stage2_batched_rounds.json "output_check"): K = 2^16 words; the 13
    program words (the ELF's 104 code bytes as little-endian u64) at index 4096.. in both val_init and val_final; no inputs, no
    outputs, panic bit 0; the termination bit 1 in val_final and val_io only; io_mask = 1 on [io_start, io_end); eq table of
    r_address with r[0] the most significant variable (:586-609). Returns (eq, io_mask, val_final, val_io, val_init) as (K,4) limbs."""
    import struct
    K = oc["K"]
    zero, one = fr_from_int(0), fr_from_int(1)
    words = struct.unpack("<%dQ" % oc["ram_words"], elf_bytes[0x1000:0x1000 + 8 * oc["ram_words"]])
    val_init = np.tile(zero, (K, 1))
    for i, w in enumerate(words):
        val_init[oc["first_ram_index"] + i] = fr_from_int(w)
    val_final = val_init.copy()
    val_final[oc["termination_index"]] = one
    val_io = np.tile(zero, (K, 1))
    val_io[oc["termination_index"]] = one
    io_mask = np.tile(zero, (K, 1))
    io_mask[oc["io_start"]:oc["io_end"]] = one
    r = np.stack([fr_from_int(int.from_bytes(bytes.fromhex(h), "little")) for h in oc["r_address"]])
    return eq_table(r), io_mask, val_final, val_io, val_init


def rwc_inputs_of_the_captured_run(rwc, stage2, elf_bytes, fr_from_int):
    """The inputs of the RamReadWriteCheckingProver of the reference's captured run (tests/golden/rwc_captured_run.json, with the
    Stage-1 / Stage-2 challenges of tests/golden/stage2_batched_rounds.json): (accesses, gamma, r_cycle, initial_ram, challenges).
    The 13 program words are the ELF's 104 code bytes at 0x80000000 (the file keeps them at offset 0x1000)."""
    import struct
    words = struct.unpack("<%dQ" % rwc["initial_ram_entries"], elf_bytes[0x1000:0x1000 + 8 * rwc["initial_ram_entries"]])
    initial_ram = {0x80000000 + 8 * i: w for i, w in enumerate(words)}
    for s in rwc["init"]["val_init_shown"]:  # the five the log shows
        assert initial_ram[s["addr"]] == s["val"] and (s["addr"] - rwc["start_address"]) // 8 == s["idx"]
    e = rwc["init"]["entries"][0]
    accesses = [(e["cycle"], rwc["start_address"] + 8 * e["addr"], e["op"] == 1, e["next_val"])]
    gamma = fr_from_int(int(rwc["gamma_be"], 16))
    r_cycle = np.stack([fr_from_int(int.from_bytes(bytes.fromhex(h), "little")) for h in stage2["stage1_r_cycle"]])
    challenges = np.stack([fr_from_int(int.from_bytes(bytes.fromhex(r["challenge"]), "little")) for r in stage2["rounds"]])
    return accesses, gamma, r_cycle, initial_ram, challenges
