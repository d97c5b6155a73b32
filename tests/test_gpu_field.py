"""Device field arithmetic (zolt_amd/csrc/field.hip.h) vs the CPU oracle, through the C ABI."""
import numpy as np
import pytest

from tests import util as U

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def zl():
    from zolt_amd import lib
    lib.init()
    return lib


def _edge_raw(mod):
    from oracle import pymodel as pm
    vals = [0, 1, 2, mod - 1, mod - 2, mod, mod + 1, (1 << 256) - 1, (1 << 256) % mod, (1 << 255), (1 << 64) - 1, 1 << 64,
            (1 << 128) - 1, ((1 << 256) % mod) - 1, mod >> 1]
    return np.array([pm.limbs(v) for v in vals], dtype=np.uint64)


@pytest.mark.parametrize("field", [0, 1])
def test_field_ops_match_oracle(zl, field):
    from oracle import binding as ob
    from oracle import pymodel as pm
    mod = pm.R_MOD if field == 0 else pm.P_MOD
    n = 1 << 20  # >= 10^6 random pairs
    a_raw = U.random_raw256(101 + field, n)
    b_raw = U.random_raw256(202 + field, n)
    e = _edge_raw(mod)
    a_raw[: len(e)] = e
    b_raw[: len(e)] = e[::-1]
    b_raw[len(e): 2 * len(e)] = e  # edge x edge on the diagonal too
    a_raw[len(e): 2 * len(e)] = e
    # TO_MONT accepts raw values >= modulus (fromBytes, src/field/mod.zig:171-184)
    a = zl.field_op(field, zl.OP_TO_MONT, a_raw)
    b = zl.field_op(field, zl.OP_TO_MONT, b_raw)
    assert np.array_equal(a, ob.f_to_mont(field, a_raw))
    assert np.array_equal(b, ob.f_to_mont(field, b_raw))
    assert np.array_equal(zl.field_op(field, zl.OP_MUL, a, b), ob.f_mul(field, a, b))
    assert np.array_equal(zl.field_op(field, zl.OP_ADD, a, b), ob.f_add(field, a, b))
    assert np.array_equal(zl.field_op(field, zl.OP_SUB, a, b), ob.f_sub(field, a, b))
    assert np.array_equal(zl.field_op(field, zl.OP_NEG, a), ob.f_neg(field, a))
    assert np.array_equal(zl.field_op(field, zl.OP_SQR, a), ob.f_sqr(field, a))
    assert np.array_equal(zl.field_op(field, zl.OP_FROM_MONT, a), ob.f_from_mont(field, a))
    k = 4096
    assert np.array_equal(zl.field_op(field, zl.OP_INV, a[:k]), ob.f_inv(field, a[:k]))
    k = 1 << 16  # binary-Euclid inversion used by the device toAffine
    assert np.array_equal(zl.field_op(field, zl.OP_INV_XGCD, a[:k]), ob.f_inv(field, a[:k]))
    assert np.array_equal(zl.field_op(field, zl.OP_INV_SAFEGCD, a[:k]), ob.f_inv(field, a[:k]))


@pytest.mark.parametrize("field", [0, 1])
def test_inversion_variants_on_structured_values(zl, field):
    """The three device inversions (Fermat, binary Euclid, batched division steps — the last is what toAffine
    uses) against the oracle's Fermat inverse on values that stress their control flow: 0 (-> 0, src/field/mod.zig:500-503),
    small integers, powers of two and their neighbours (long runs of trailing zeros), modulus - small, all-ones limbs, and
    the Montgomery images of the same."""
    from oracle import binding as ob
    mod = 21888242871839275222246405745257275088548364400416034343698204186575808495617 if field == 0 else \
        21888242871839275222246405745257275088696311157297823662689037894645226208583
    vals = [0, 1, 2, 3, 4, 7, 255, 256, mod - 1, mod - 2, mod - 3, (mod - 1) // 2, (mod + 1) // 2, (1 << 253) % mod, (1 << 253) - 1,
            (1 << 254) % mod, (1 << 255) % mod, (1 << 128), (1 << 128) - 1, (1 << 192) + 1, (1 << 64), (1 << 32), (1 << 30),
            (1 << 60) - 1, 0xFFFFFFFF, 0xFFFFFFFFFFFFFFFF, ((1 << 256) - 1) % mod]
    vals += [(1 << k) % mod for k in range(1, 254, 7)] + [((1 << k) + 1) % mod for k in range(29, 254, 30)]
    raw = np.array([[(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)] for v in vals], dtype=np.uint64)
    for a in (raw, ob.f_to_mont(field, raw)):   # every 256-bit pattern below the modulus is a valid Montgomery value
        want = ob.f_inv(field, a)
        for op in (zl.OP_INV, zl.OP_INV_XGCD, zl.OP_INV_SAFEGCD):
            assert np.array_equal(zl.field_op(field, op, a), want), op
        # x * x^-1 == 1 (Montgomery one) wherever x != 0
        one = ob.f_from_u64(field, np.array([1], dtype=np.uint64))[0]
        prod = ob.f_mul(field, a, want)
        for i, v in enumerate(a):
            assert np.array_equal(prod[i], one) or not v.any()


def test_field_kats(zl):
    """src/field/mod.zig:1101-1140: 3*7 = 21, 7*7^-1 = 1, 2^3 = 8."""
    f = U.fr([3, 7, 2, 21, 8, 1])
    assert np.array_equal(zl.field_op(0, zl.OP_MUL, f[0:1], f[1:2])[0], f[3])
    inv7 = zl.field_op(0, zl.OP_INV, f[1:2])
    assert np.array_equal(zl.field_op(0, zl.OP_MUL, f[1:2], inv7)[0], f[5])
    sq = zl.field_op(0, zl.OP_SQR, f[2:3])
    assert np.array_equal(zl.field_op(0, zl.OP_MUL, sq, f[2:3])[0], f[4])
    z = np.zeros((1, 4), dtype=np.uint64)
    assert not zl.field_op(0, zl.OP_NEG, z).any() and not zl.field_op(0, zl.OP_INV, z).any()


def test_invalid_arguments(zl):
    with pytest.raises(zl.ZgError):
        zl.field_op(7, zl.OP_MUL, np.zeros((1, 4), dtype=np.uint64), np.zeros((1, 4), dtype=np.uint64))


def test_lazy_29bit_limb_arithmetic(zl):
    """csrc/fp29.hip.h (the MSM inner loop's representation): products, squares and a chain through every
    biased subtraction and the exact zero test, on 2^20 random pairs plus edge values, against big ints."""
    from oracle import binding as ob
    from oracle import pymodel as pm
    p = pm.P_MOD
    n = 1 << 20
    a_raw = U.random_raw256(909, n)
    b_raw = U.random_raw256(910, n)
    e = _edge_raw(p)
    a_raw[: len(e)] = e
    b_raw[: len(e)] = e[::-1]
    a_raw[len(e): 2 * len(e)] = e
    b_raw[len(e): 2 * len(e)] = e
    a = ob.f_to_mont(ob.FP, a_raw)
    b = ob.f_to_mont(ob.FP, b_raw)
    assert np.array_equal(zl.field_op(1, zl.OP_MUL29, a, b), ob.f_mul(ob.FP, a, b))
    assert np.array_equal(zl.field_op(1, zl.OP_SQR29, a, b), ob.f_sqr(ob.FP, a))
    # chain: b^2 + b^3 - 8ab (see fp29_op_kernel)
    b2 = ob.f_sqr(ob.FP, b)
    b3 = ob.f_mul(ob.FP, b2, b)
    ab = ob.f_mul(ob.FP, a, b)
    eight = ob.f_from_u64(ob.FP, np.full(n, 8, dtype=np.uint64))
    want = ob.f_sub(ob.FP, ob.f_add(ob.FP, b2, b3), ob.f_mul(ob.FP, eight, ab))
    assert np.array_equal(zl.field_op(1, zl.OP_X3_29, a, b), want)
    with pytest.raises(zl.ZgError):
        zl.field_op(0, zl.OP_MUL29, a[:4], b[:4])  # Fp only
