"""witness.py — the R1CS cycle inputs as INTEGER COLUMNS, widened to field elements on the device.

The reference builds a 43-element row of field elements per cycle on the CPU (R1CSWitnessGenerator.generateWitness,
src/zkvm/r1cs/constraints.zig:1469-1494: createNoopWitness :1418-1438 for padding, fromTraceStep :929-1223 for a real step) and every
prover that follows reads that matrix. Each of those 43 values is F.fromU64 of a machine integer, signedI64ToField of an immediate, a
0/1 flag, or a sum / product of two such values — so the same row exists as ~156 bytes of integers, and `zg_fr_rows_from_columns`
(include/zolt_gpu.h) produces the identical 1376-byte row in HBM. cycleColumnsFromTrace is the integer-domain restatement of the
generator (what a Zig shim computes instead of the field-element rows); CycleWitnessMatrix is the one device-resident matrix that
Stage 1 (StreamingOuterProver, R1CSInputEvaluator), Stage 2 (product virtualisation) and Stage 3 share.

Part of the zolt_amd.api package; import zolt_amd.api, which re-exports every name of every part."""
import numpy as np

from .. import lib
from ._base import *  # noqa: F401,F403

R1CS_INPUT_NAMES = (
    "LeftInstructionInput", "RightInstructionInput", "Product", "WriteLookupOutputToRD", "WritePCtoRD", "ShouldBranch", "PC", "UnexpandedPC", "Imm",
    "RamAddress", "Rs1Value", "Rs2Value", "RdWriteValue", "RamReadValue", "RamWriteValue", "LeftLookupOperand", "RightLookupOperand",
    "NextUnexpandedPC", "NextPC", "NextIsVirtual", "NextIsFirstInSequence", "LookupOutput", "ShouldJump", "FlagAddOperands",
    "FlagSubtractOperands", "FlagMultiplyOperands", "FlagLoad", "FlagStore", "FlagJump", "FlagWriteLookupOutputToRD", "FlagVirtualInstruction",
    "FlagAssert", "FlagDoNotUpdateUnexpandedPC", "FlagAdvice", "FlagIsCompressed", "FlagIsFirstInSequence", "FlagIsRdNotZero", "FlagBranch",
    "FlagIsNoop", "FlagLeftOperandIsRs1", "FlagLeftOperandIsPC", "FlagRightOperandIsRs2", "FlagRightOperandIsImm")  # R1CSInputIndex (:38-86)
_W = {n: i for i, n in enumerate(R1CS_INPUT_NAMES)}
# the inputs that are single bits: they travel as bits of ONE 32-bit word per cycle
_BIT_INPUTS = ("WriteLookupOutputToRD", "WritePCtoRD", "ShouldBranch", "ShouldJump") + tuple(n for n in R1CS_INPUT_NAMES if n.startswith("Flag"))
_M64 = (1 << 64) - 1


def _sx(v, bits):
    return v - (1 << bits) if v >> (bits - 1) else v


def _imm_of(w):
    """deriveImmediate (:1226-1274) as a signed integer: I / S / B / J sign-extended, U-type the unsigned upper bits, everything else 0"""
    op = w & 0x7F
    if op in (0x13, 0x03, 0x67):
        return _sx(w >> 20, 12)
    if op == 0x23:
        return _sx((((w >> 25) & 0x7F) << 5) | ((w >> 7) & 0x1F), 12)
    if op == 0x63:
        return _sx((((w >> 31) & 1) << 12) | (((w >> 7) & 1) << 11) | (((w >> 25) & 0x3F) << 5) | (((w >> 8) & 0xF) << 1), 13)
    if op == 0x6F:
        return _sx((((w >> 31) & 1) << 20) | (((w >> 12) & 0xFF) << 12) | (((w >> 20) & 1) << 11) | (((w >> 21) & 0x3FF) << 1), 21)
    if op in (0x37, 0x17):
        return w & 0xFFFFF000
    return 0


def _next_is_noop(step):
    """isNoopInstruction (:569-595): a padding cycle, or the canonical ADDI x0, x0, 0"""
    if step is None:
        return False
    if step["is_noop"]:
        return True
    w = step["instruction"]
    return (w & 0x7F) == 0x13 and ((w >> 7) & 31) == 0 and ((w >> 15) & 31) == 0 and ((w >> 12) & 7) == 0 and (w >> 20) == 0


def _cycle_integers(step, nxt):
    """one real cycle (fromTraceStep): {input name: exact integer (may be negative)} and the set of flag names that are 1"""
    w = step["instruction"]
    op, f3, f7, rd = w & 0x7F, (w >> 12) & 7, (w >> 25) & 0x7F, (w >> 7) & 31
    v, bits = {}, set()
    load, store, branch = op == 0x03, op == 0x23, op == 0x63
    if load:
        bits.add("FlagLoad")
    if store:
        bits.add("FlagStore")
    if step["is_compressed"]:
        bits.add("FlagIsCompressed")
    imm = _imm_of(w)
    v["Imm"] = imm
    rs1 = step["rs1_value"] if op in (0x13, 0x03, 0x67, 0x1B, 0x33, 0x3B, 0x23, 0x63) else 0  # :957-977
    rs2 = step["rs2_value"] if op in (0x33, 0x3B, 0x23, 0x63) else 0                          # :986-993
    v["Rs1Value"], v["Rs2Value"] = rs1, rs2
    v["RamAddress"] = step["rs1_value"] + imm if (load or store) else 0                         # :1001-1009, in the field: no wrap
    mem = step["memory_value"] or 0
    if load:                                                                                    # :1023-1047
        v["RamReadValue"] = v["RamWriteValue"] = v["RdWriteValue"] = mem
    elif store:
        v["RamReadValue"], v["RamWriteValue"] = mem, step["rs2_value"]
    elif not branch and rd != 0:
        v["RdWriteValue"] = step["rd_value"]
    l_rs1, l_pc = op in (0x33, 0x13, 0x03, 0x67, 0x23, 0x63, 0x1B, 0x3B), op in (0x17, 0x6F)    # :1059-1098
    r_rs2, r_imm = op in (0x33, 0x63, 0x3B), op in (0x13, 0x03, 0x67, 0x23, 0x37, 0x17, 0x6F, 0x1B)
    for name, on in (("FlagLeftOperandIsRs1", l_rs1), ("FlagLeftOperandIsPC", l_pc), ("FlagRightOperandIsRs2", r_rs2), ("FlagRightOperandIsImm", r_imm)):
        if on:
            bits.add(name)
    left = (rs1 if l_rs1 else 0) + (step["unexpanded_pc"] if l_pc else 0)                       # :1106-1118
    right = (rs2 if r_rs2 else 0) + (imm if r_imm else 0)
    v["LeftInstructionInput"], v["RightInstructionInput"] = left, right                          # Product = left * right: formed on the device
    if op == 0x6F:                                                                               # computeLookupOutput (:600-640)
        lookup = (step["pc"] + imm) & _M64
    elif op == 0x67:
        lookup = ((step["rs1_value"] + _sx(w >> 20, 12)) & _M64) & ~1
    elif branch:
        a, b = step["rs1_value"], step["rs2_value"]
        sa, sb = _sx(a, 64), _sx(b, 64)
        lookup = int({0: a == b, 1: a != b, 4: sa < sb, 5: sa >= sb, 6: a < b, 7: a >= b}.get(f3, False))
    else:
        lookup = step["rd_value"]
    v["LookupOutput"] = lookup
    v["PC"], v["UnexpandedPC"] = step["pc"], step["unexpanded_pc"]
    if nxt is not None and not nxt["is_noop"]:                                                   # :1150-1172
        v["NextPC"], v["NextUnexpandedPC"] = nxt["pc"], nxt["unexpanded_pc"]
    # setFlagsFromInstruction (:1288-1398): the circuit flags and the two lookup operands
    lo_l, lo_r, wl, jump = left, right, False, False
    if op == 0x33:
        if f7 == 0x01:
            if f3 == 0:
                bits.add("FlagMultiplyOperands")
                lo_l, lo_r = 0, 0  # = Product: added on the device (cycleColumnsFromTrace)
        elif f7 == 0x20 and f3 == 0:
            bits.add("FlagSubtractOperands")
            lo_l, lo_r = 0, left - right + (1 << 64)
        else:
            bits.add("FlagAddOperands")
            lo_l, lo_r = 0, left + right
        wl = True
    elif op == 0x13 or op in (0x37, 0x17):
        bits.add("FlagAddOperands")
        lo_l, lo_r, wl = 0, left + right, True
    elif op in (0x6F, 0x67):
        bits.add("FlagAddOperands")
        lo_l, lo_r, jump = 0, left + right, True
    v["LeftLookupOperand"], v["RightLookupOperand"] = lo_l, lo_r
    if wl:
        bits.add("FlagWriteLookupOutputToRD")
    if jump:
        bits.add("FlagJump")
        if not _next_is_noop(nxt):
            bits.add("ShouldJump")                                                                  # :1181-1185
    if rd != 0:                                                                                  # :1190-1215
        bits.add("FlagIsRdNotZero")
        if wl:
            bits.add("WriteLookupOutputToRD")
        if jump:
            bits.add("WritePCtoRD")
    if branch:
        bits.add("FlagBranch")
        if lookup:
            bits.add("ShouldBranch")
    return v, bits


def cycleColumnsFromTrace(steps):
    """ExecutionTrace.steps (dicts with tracer.TraceStep's fields, NoOp-padded) -> the 43 typed columns of zg_fr_rows_from_columns, in
    R1CSInputIndex order: a list of (kind, data, a, b). Unsigned machine words travel as u64, the immediate as i64, the three values that
    can leave 64 bits (RightInstructionInput = rs2 or a signed immediate, RamAddress = rs1 + imm, RightLookupOperand = a sum, a
    difference + 2^64 or a pass-through) as 128-bit two's complement, every single-bit input as a bit of one u32 word, Product as the
    device-side product of columns 0 and 1, RightLookupOperand's MUL rows as Product * FlagMultiplyOperands on the device (a full-width
    product fits no 128-bit signed word) added to its column, and the two always-zero inputs as no data at all: 156 bytes per cycle."""
    n = len(steps)
    u64_names = ("LeftInstructionInput", "PC", "UnexpandedPC", "Rs1Value", "Rs2Value", "RdWriteValue", "RamReadValue", "RamWriteValue", "LeftLookupOperand",
                 "NextUnexpandedPC", "NextPC", "LookupOutput")
    wide_names = ("RightInstructionInput", "RamAddress", "RightLookupOperand")
    u64 = {name: np.zeros(n, dtype=np.uint64) for name in u64_names}
    wide = {name: [0] * n for name in wide_names}
    imm = np.zeros(n, dtype=np.int64)
    word = np.zeros(n, dtype=np.uint32)
    bit_of = {name: i for i, name in enumerate(_BIT_INPUTS)}
    for i, st in enumerate(steps):
        if st["is_noop"]:                                                                        # createNoopWitness (:1418-1438)
            word[i] = (1 << bit_of["FlagDoNotUpdateUnexpandedPC"]) | (1 << bit_of["FlagIsNoop"])
            continue
        v, bits = _cycle_integers(st, steps[i + 1] if i + 1 < n else None)
        for name in u64_names:
            x = v.get(name, 0)
            assert 0 <= x <= _M64, (name, x)
            u64[name][i] = x
        for name in wide_names:
            wide[name][i] = v.get(name, 0)
        imm[i] = v["Imm"]
        w = 0
        for name in bits:
            w |= 1 << bit_of[name]
        word[i] = w
    cols = [None] * len(R1CS_INPUT_NAMES)
    for name in u64_names:
        cols[_W[name]] = (lib.COL_U64, u64[name])
    cols[_W["Imm"]] = (lib.COL_I64, imm)
    for name in wide_names:
        vals = wide[name]
        assert all(-(1 << 127) <= x < (1 << 127) for x in vals), name
        a = np.zeros((n, 2), dtype=np.uint64)
        for i, x in enumerate(vals):
            x &= (1 << 128) - 1
            a[i, 0], a[i, 1] = x & _M64, x >> 64
        # RightLookupOperand: the rows of a MUL instruction take Product (constraint 9; a full-width product does not fit 128-bit two's
        # complement) — Product * FlagMultiplyOperands on the device — and every other row its sum / difference / pass-through as the addend
        cols[_W[name]] = (lib.COL_MUL, a, _W["Product"], _W["FlagMultiplyOperands"]) if name == "RightLookupOperand" else (lib.COL_I128, a)
    cols[_W["Product"]] = (lib.COL_MUL, None, _W["LeftInstructionInput"], _W["RightInstructionInput"])
    for name in ("NextIsVirtual", "NextIsFirstInSequence"):  # no virtual sequences in a RISC-V trace (:1160-1171)
        cols[_W[name]] = (lib.COL_ZERO, None)
    for name in _BIT_INPUTS:
        cols[_W[name]] = (lib.COL_BIT, word, bit_of[name], 4)
    assert all(c is not None for c in cols)
    return cols


def columnBytesPerCycle(cols):
    """bytes of column data that cross PCIe per cycle (a shared flag word counted once)"""
    seen, total = set(), 0
    width = {lib.COL_U8: 1, lib.COL_U32: 4, lib.COL_U64: 8, lib.COL_I64: 8, lib.COL_I128: 16, lib.COL_U128: 16, lib.COL_FR: 32, lib.COL_MUL: 16}
    for c in cols:
        if len(c) < 2 or c[1] is None or id(c[1]) in seen:
            continue
        seen.add(id(c[1]))
        total += c[3] if c[0] == lib.COL_BIT else width[c[0]]
    return total


class CycleWitnessMatrix:
    """The cycle-major witness matrix (num_cycles x 43 field elements, src/zkvm/r1cs/evaluation.zig:55-122) resident in HBM, built once
    and shared by the stages that read it. `from_columns` widens integer columns on the device; `from_witnesses` uploads ready rows."""

    def __init__(self, buf, num_cycles):
        self._buf, self.num_cycles = buf, num_cycles

    @classmethod
    def from_columns(cls, cols, num_cycles):
        buf = lib.DeviceBuffer(max(num_cycles * len(cols) * 32, 32))
        lib.fr_rows_from_columns(cols, num_cycles, buf.ptr)
        return cls(buf, num_cycles)

    @classmethod
    def from_trace(cls, steps):
        return cls.from_columns(cycleColumnsFromTrace(steps), len(steps))

    @classmethod
    def from_witnesses(cls, cycle_witnesses):
        w = np.ascontiguousarray(cycle_witnesses, dtype=np.uint64).reshape(-1, len(R1CS_INPUT_NAMES), 4)
        return cls(lib.DeviceBuffer.from_host(w), w.shape[0])

    @property
    def ptr(self):
        return self._buf.ptr

    def to_host(self):
        return self._buf.to_host()[:self.num_cycles * len(R1CS_INPUT_NAMES) * 4].reshape(self.num_cycles, len(R1CS_INPUT_NAMES), 4)

    def free(self):
        self._buf.free()
