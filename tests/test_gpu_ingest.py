"""zg_fr_rows_from_columns on the device (csrc/ingest.hip): integer columns -> the cycle-major matrix of Montgomery elements, bit for bit
what the reference's conversions produce (F.fromU64, signedI64ToField src/zkvm/r1cs/constraints.zig:868-876, flags, field products),
checked against the big-integer model of tests/test_witness_columns.py and, through it, against the oracle's restatement of
R1CSWitnessGenerator.generateWitness on the captured fibonacci run and on random traces."""
import os

import numpy as np
import pytest

from oracle import binding as ob  # checker
from tests import util as U
from tests.test_witness_columns import oracle_rows_int, random_trace, widen_columns_model
from zolt_amd import api, lib

pytestmark = pytest.mark.gpu


def rows_int(m):
    return [[api.fr_to_int(x) for x in row] for row in m]


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 4097])
def test_every_column_kind_at_the_edges(n):
    lib.init(0)
    rng = np.random.default_rng(n)
    edge64 = np.array([0, 1, 2, (1 << 63) - 1, 1 << 63, (1 << 64) - 1], dtype=np.uint64)
    u64 = rng.integers(0, 1 << 63, size=n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=n, dtype=np.uint64)
    u64[:min(n, 6)] = edge64[:min(n, 6)]
    i64 = u64.view(np.int64).copy()  # INT64_MIN, -1, ... among the edges
    wide = np.stack([rng.permutation(u64), rng.integers(0, 1 << 63, size=n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=n, dtype=np.uint64)], axis=1)
    edges128 = [(0, 0), (1, 0), ((1 << 64) - 1, (1 << 64) - 1), (0, 1 << 63), ((1 << 64) - 1, (1 << 63) - 1), (0, 1)]
    for k, (lo, hi) in enumerate(edges128[:n]):
        wide[k] = (lo, hi)
    frs = lib.field_op(lib.FR, lib.OP_TO_MONT, rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64))
    word32 = rng.integers(0, 1 << 32, size=n, dtype=np.uint32)
    word64 = rng.integers(0, 1 << 63, size=n, dtype=np.uint64)
    word8 = rng.integers(0, 256, size=n, dtype=np.uint8)
    cols = [(lib.COL_U64, u64), (lib.COL_I64, i64), (lib.COL_MUL, None, 0, 1), (lib.COL_U8, word8), (lib.COL_U32, word32), (lib.COL_I128, wide),
            (lib.COL_U128, wide), (lib.COL_FR, frs), (lib.COL_ZERO, None), (lib.COL_BIT, word32, 0, 4), (lib.COL_BIT, word32, 31, 4),
            (lib.COL_BIT, word64, 62, 8), (lib.COL_BIT, word8, 7, 1), (lib.COL_MUL, None, 5, 7), (lib.COL_MUL, None, 6, 6),
            (lib.COL_MUL, wide, 2, 9), (lib.COL_MUL, wide[::-1].copy(), 0, 12),  # product of a product + an addend; product + an addend
            (lib.COL_LUT, word8, 1, 200, frs[:200] if n >= 200 else np.concatenate([frs] * (200 // n + 1))[:200]),  # indices >= 200 read as zero
            (lib.COL_LUT, (word32 & 0x3FF).astype(np.uint16), 2, 1000, np.concatenate([frs] * (1000 // n + 1))[:1000]),
            (lib.COL_LUT, word32 & np.uint32(31), 4, 32, np.concatenate([frs] * (32 // n + 1))[:32])]
    got = lib.fr_rows_from_columns(cols, n)
    assert got.shape == (n, len(cols), 4)
    assert rows_int(got) == widen_columns_model(cols, n)
    assert np.all(api.fr_to_int(got[i, 8]) == 0 for i in range(n))
    # the same from columns that are already in HBM
    bufs, dcols = [], []
    for spec in cols:
        if len(spec) > 1 and spec[1] is not None:
            bufs.append(lib.DeviceBuffer.from_host(spec[1]))
            d = (spec[0], bufs[-1].ptr) + tuple(spec[2:4])
            if len(spec) > 4:
                bufs.append(lib.DeviceBuffer.from_host(spec[4]))
                d += (bufs[-1].ptr,)
            dcols.append(d)
        else:
            dcols.append(spec)
    out = lib.DeviceBuffer(n * len(cols) * 32)
    lib.fr_rows_from_columns_dev(dcols, n, out.ptr)
    lib.sync()
    assert np.array_equal(out.to_host()[:n * len(cols) * 4].reshape(n, len(cols), 4), got)
    for b in bufs + [out]:
        b.free()


def test_invalid_column_descriptions_are_refused():
    lib.init(0)
    out = lib.DeviceBuffer(64 * 32)
    a = np.zeros(4, dtype=np.uint64)
    p = a.ctypes.data
    bad = [[(lib.COL_MUL, 0, 0, None)],                                   # a product of itself
           [(lib.COL_U64, 0, 0, p), (lib.COL_MUL, 0, 5, None)],           # a factor outside the matrix
           [(lib.COL_U64, 0, 0, p), (lib.COL_MUL, 0, 0, None), (lib.COL_MUL, 1, 0, None), (lib.COL_MUL, 2, 0, None)],  # nested three deep
           [(lib.COL_BIT, 64, 8, p)], [(lib.COL_BIT, 3, 2, p)],           # a bit outside its word, a word width that does not exist
           [(lib.COL_BIT, 3, 4, None)], [(lib.COL_U64, 0, 0, None)],      # no data
           [(17, 0, 0, p)]]
    for cols in bad:
        arr = (lib.Column * len(cols))(*[lib.Column(k, x, y, d) for k, x, y, d in cols])
        rc = lib._lib.zg_fr_rows_from_columns(arr, len(cols), 4, out.ptr)
        assert rc == lib.ERR_INVALID, cols
    with pytest.raises(lib.ZgError):
        lib.fr_rows_from_columns([(lib.COL_ZERO, None)] * 65, 4, out.ptr)
    lib.fr_rows_from_columns([(lib.COL_ZERO, None)] * 64, 0, out.ptr)  # no rows: nothing to do
    out.free()


def test_witness_matrix_of_the_captured_run_from_trace_columns(golden_dir):
    """the 256 x 43 matrix of `zolt prove examples/fibonacci.elf` built on the device from 156 bytes of integers per cycle == the rows the
    reference's generator produces (oracle restatement), and the provers that read it give the same answers from either form"""
    lib.init(0)
    elf = open(os.path.join(golden_dir, "fibonacci.elf"), "rb").read()
    steps = U.fibonacci_full_trace(elf)
    want = ob.r1cs_witness_from_trace(steps)
    m = api.CycleWitnessMatrix.from_trace(steps)
    assert np.array_equal(m.to_host(), want)
    rng = np.random.default_rng(5)
    r = lib.field_op(lib.FR, lib.OP_TO_MONT, rng.integers(0, 1 << 62, size=(8, 4), dtype=np.uint64))
    assert np.array_equal(api.R1CSInputEvaluator.computeClaimedInputs(m, r), api.R1CSInputEvaluator.computeClaimedInputs(want, r))
    tau = lib.field_op(lib.FR, lib.OP_TO_MONT, rng.integers(0, 1 << 62, size=(10, 4), dtype=np.uint64))
    p_cols, p_rows = api.StreamingOuterProver(m, tau), api.StreamingOuterProver(want, tau)
    assert np.array_equal(p_cols.computeFirstRoundPoly(), p_rows.computeFirstRoundPoly())
    p_cols.deinit()
    p_rows.deinit()
    assert np.array_equal(m.to_host(), want)  # a prover that borrowed the shared matrix does not free it
    m.free()


@pytest.mark.parametrize("seed,n", [(11, 3000), (12, 70000)])
def test_witness_matrix_of_random_traces(seed, n):
    lib.init(0)
    steps = random_trace(seed, n, 37)
    cols = api.cycleColumnsFromTrace(steps)
    m = api.CycleWitnessMatrix.from_columns(cols, len(steps))
    got = m.to_host()
    m.free()
    if n <= 5000:
        assert rows_int(got) == oracle_rows_int(steps)
    else:  # the oracle's per-element Python conversion is too slow here: the big-integer model on a sample of rows, a checksum on all
        model = widen_columns_model([(c[0],) + ((c[1][::997],) if len(c) > 1 and c[1] is not None else (None,)) + tuple(c[2:]) for c in cols], len(steps[::997]))
        assert rows_int(got[::997]) == model
        again = lib.fr_rows_from_columns(cols, len(steps))
        assert np.array_equal(again, got)


def test_session_from_one_column():
    """zg_sumcheck_open_column: a session whose table is one widened column, zero past n_rows — the tables of proveStage5 (a 32-entry lookup
    by the rd byte of every cycle) and proveStage6 (all zero) without a 32-byte-per-entry upload; rounds equal a session opened on the
    host-built table"""
    lib.init(0)
    rng = np.random.default_rng(3)
    table = lib.field_op(lib.FR, lib.OP_TO_MONT, rng.integers(0, 1 << 62, size=(32, 4), dtype=np.uint64))
    for n_rows, n in ((1000, 1024), (1, 2), (4096, 4096), (0, 64)):
        rd = rng.integers(0, 32, size=n_rows, dtype=np.uint8)
        want = np.zeros((n, 4), dtype=np.uint64)
        want[:n_rows] = table[rd]
        col = (lib.COL_LUT, rd, 1, 32, table) if n_rows else (lib.COL_ZERO, None)
        s1, s2 = lib.SumcheckSession.open_column(col, n_rows, n, lib.SC_HIGH_HALF), lib.SumcheckSession.open(want, lib.SC_HIGH_HALF)
        assert np.array_equal(s1.read(), want)
        while len(s1) > 1:
            a, b = s1.round_sums(), s2.round_sums()
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
            r = table[len(s1) % 32]
            s1.bind(r)
            s2.bind(r)
        assert np.array_equal(s1.final(), s2.final())
        s1.close()
        s2.close()
    with pytest.raises(lib.ZgError):
        lib.SumcheckSession.open_column((lib.COL_ZERO, None), 5, 4)
