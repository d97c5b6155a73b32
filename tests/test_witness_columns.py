"""The R1CS cycle inputs as integer columns (zolt_amd/api/witness.py: what crosses PCIe instead of 1376-byte rows of field elements).

CPU: cycleColumnsFromTrace — the product's integer-domain restatement of R1CSWitnessGenerator.generateWitness
(src/zkvm/r1cs/constraints.zig:929-1223, 1418-1438, 1469-1494) — widened by a big-integer model of zg_fr_rows_from_columns must equal
the oracle's restatement of the same generator, on the captured fibonacci run and on random traces that reach every opcode branch,
negative immediates, 128-bit products and the wrap cases. GPU (tests/test_gpu_ingest.py): the device kernel against the same model."""
import numpy as np
import pytest

from oracle import binding as ob  # checker
from tests import util as U
from zolt_amd import api, lib

R = ob.FR_MOD if hasattr(ob, "FR_MOD") else api.R_MOD


def widen_columns_model(cols, n):
    """big-integer model of zg_fr_rows_from_columns (include/zolt_gpu.h): -> n x len(cols) integers mod r"""
    out = [[0] * len(cols) for _ in range(n)]
    for c, spec in enumerate(cols):
        kind, data = spec[0], spec[1] if len(spec) > 1 else None
        a, b = (spec[2] if len(spec) > 2 else 0), (spec[3] if len(spec) > 3 else 0)
        for i in range(n):
            if kind in (lib.COL_U8, lib.COL_U32, lib.COL_U64, lib.COL_I64):
                v = int(data[i])
            elif kind in (lib.COL_I128, lib.COL_U128):
                v = int(data[i][0]) | (int(data[i][1]) << 64)
                if kind == lib.COL_I128 and v >> 127:
                    v -= 1 << 128
            elif kind == lib.COL_FR:
                v = api.fr_to_int(data[i])
            elif kind == lib.COL_BIT:
                v = (int(data[i]) >> a) & 1
            elif kind == lib.COL_LUT:
                ix = int(data[i])
                v = api.fr_to_int(spec[4][ix]) if ix < b else 0
            else:
                continue
            out[i][c] = v % R
    def is_mul(k):
        return cols[k][0] == lib.COL_MUL
    for depth in (1, 2):  # a factor may be a depth-1 product; the optional data is a 128-bit two's-complement addend
        for c, spec in enumerate(cols):
            if spec[0] != lib.COL_MUL or (2 if (is_mul(spec[2]) or is_mul(spec[3])) else 1) != depth:
                continue
            for i in range(n):
                v = out[i][spec[2]] * out[i][spec[3]]
                if spec[1] is not None:
                    a = int(spec[1][i][0]) | (int(spec[1][i][1]) << 64)
                    v += a - (1 << 128) if a >> 127 else a
                out[i][c] = v % R
    return out


def random_trace(seed, n, pad):
    """n real steps over every opcode class fromTraceStep distinguishes (+ unknown ones), operands at the edges, then NoOp padding"""
    rng = np.random.default_rng(seed)
    edge = [0, 1, 2, (1 << 63) - 1, 1 << 63, (1 << 64) - 1, (1 << 64) - 2, 0x80000000, 0xFFFFFFFF]
    ops = [0x33, 0x13, 0x03, 0x23, 0x63, 0x37, 0x17, 0x6F, 0x67, 0x1B, 0x3B, 0x73, 0x0F]

    def val():
        return int(edge[rng.integers(len(edge))]) if rng.random() < 0.4 else int(rng.integers(0, 1 << 63)) * 2 + int(rng.integers(0, 2))

    steps = []
    for _ in range(n):
        op = ops[rng.integers(len(ops))]
        w = int(rng.integers(0, 1 << 32)) & ~0x7F | op
        if op == 0x33:  # make MUL / SUB / ADD all likely
            f7 = [0x01, 0x20, 0x00, int(rng.integers(0, 128))][rng.integers(4)]
            f3 = [0, 0, int(rng.integers(0, 8))][rng.integers(3)]
            w = (w & ~((0x7F << 25) | (7 << 12))) | (f7 << 25) | (f3 << 12)
        if rng.random() < 0.2:
            w &= ~(31 << 7)  # rd = x0
        if rng.random() < 0.05:
            w = 0x13  # the canonical NOP (isNoopInstruction)
        pc = int(rng.integers(0, 1 << 20)) * 4 + 0x80000000
        steps.append({"instruction": w, "pc": pc, "unexpanded_pc": pc if rng.random() < 0.7 else val(), "rs1_value": val(), "rs2_value": val(), "rd_value": val(),
                      "memory_value": None if rng.random() < 0.3 else val(), "is_compressed": bool(rng.random() < 0.2), "is_noop": False})
    noop = {"instruction": 0, "pc": 0, "unexpanded_pc": 0, "rs1_value": 0, "rs2_value": 0, "rd_value": 0, "memory_value": None, "is_compressed": False, "is_noop": True}
    return steps + [dict(noop) for _ in range(pad)]


def oracle_rows_int(steps):
    return [[api.fr_to_int(x) for x in row] for row in ob.r1cs_witness_from_trace(steps)]


def test_columns_of_the_captured_fibonacci_run_widen_to_the_reference_witness(golden_dir):
    import os
    elf = open(os.path.join(golden_dir, "fibonacci.elf"), "rb").read()
    steps = U.fibonacci_full_trace(elf)
    cols = api.cycleColumnsFromTrace(steps)
    assert len(cols) == 43 and api.columnBytesPerCycle(cols) == 156  # against 43 * 32 = 1376 bytes of field elements per cycle
    assert widen_columns_model(cols, len(steps)) == oracle_rows_int(steps)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_columns_of_random_traces_widen_to_the_reference_witness(seed):
    steps = random_trace(seed, 400, 7)
    cols = api.cycleColumnsFromTrace(steps)
    got, want = widen_columns_model(cols, len(steps)), oracle_rows_int(steps)
    for i, (g, w) in enumerate(zip(got, want)):
        assert g == w, (i, hex(steps[i]["instruction"]), [api.R1CS_INPUT_NAMES[k] for k in range(43) if g[k] != w[k]])
    # the wide columns are 128-bit two's complement; RightLookupOperand is Product * FlagMultiplyOperands + a 128-bit addend
    kinds = {api.R1CS_INPUT_NAMES[k]: cols[k][0] for k in range(43)}
    assert kinds["RightInstructionInput"] == lib.COL_I128 and kinds["RamAddress"] == lib.COL_I128 and kinds["RightLookupOperand"] == lib.COL_MUL
    assert api.columnBytesPerCycle(cols) == 156  # whatever the trace holds (a full-width MUL takes no wider encoding)



def test_a_trace_that_ends_on_a_real_step_has_no_successor():
    steps = random_trace(9, 17, 0)
    assert widen_columns_model(api.cycleColumnsFromTrace(steps), 17) == oracle_rows_int(steps)


def test_a_full_width_mul_beside_a_negative_row_keeps_the_narrow_encoding():
    """MUL x1, x2, x3 with rs1 = rs2 = 2^64 - 1: Product = RightLookupOperand = (2^64 - 1)^2 >= 2^127 fits no signed 128-bit word, and
    an ADDI with a negative immediate beside it needs the sign — the MUL rows ride on the device-side Product instead of a wider column"""
    base = {"pc": 0x80000000, "unexpanded_pc": 0x80000000, "rd_value": 1, "memory_value": None, "is_compressed": False, "is_noop": False}
    mul = dict(base, instruction=(1 << 25) | (3 << 20) | (2 << 15) | (1 << 7) | 0x33, rs1_value=(1 << 64) - 1, rs2_value=(1 << 64) - 1)
    addi = dict(base, instruction=(0xFFF << 20) | (2 << 15) | (1 << 7) | 0x13, rs1_value=0, rs2_value=0)  # ADDI x1, x2, -1 with rs1 = 0: -1
    steps = [mul, addi, mul]
    cols = api.cycleColumnsFromTrace(steps)
    assert api.columnBytesPerCycle(cols) == 156
    got, want = widen_columns_model(cols, 3), oracle_rows_int(steps)
    assert got == want and want[0][16] == ((1 << 64) - 1) ** 2 % R and want[1][16] == R - 1 and want[0][2] == want[0][16]
