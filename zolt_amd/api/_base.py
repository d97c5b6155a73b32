"""_base.py — host scalar helpers, constants (representation only).

Part of the zolt_amd.api package (the host mirror of the reference's module API over libzolt_gpu.so); import zolt_amd.api,
which re-exports every name of every part."""
import numpy as np

from .. import lib

R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617
P_MOD = 21888242871839275222246405745257275088696311157297823662689037894645226208583
_M64 = (1 << 64) - 1
_MONT_R = 1 << 256
_RINV_R = pow(_MONT_R, -1, R_MOD)
_RINV_P = pow(_MONT_R, -1, P_MOD)


# ---- host scalar helpers (representation only)
def _limbs(v):
    return np.array([(v >> (64 * i)) & _M64 for i in range(4)], dtype=np.uint64)


def _int(l):
    return sum(int(x) << (64 * i) for i, x in enumerate(l))


def fr_from_int(v):
    """F.fromU64 / canonical integer -> Montgomery limbs (src/field/mod.zig:617-622)."""
    return _limbs((v % R_MOD) * _MONT_R % R_MOD)


def fr_to_int(l):
    return _int(l) * _RINV_R % R_MOD


def fp_from_int(v):
    return _limbs((v % P_MOD) * _MONT_R % P_MOD)


def fp_to_int(l):
    return _int(l) * _RINV_P % P_MOD


def generator():
    """AffinePoint.generator() = (1, 2) (src/msm/mod.zig:43-49)."""
    return np.concatenate([fp_from_int(1), fp_from_int(2)])


def commitment_to_bytes(xy, inf):
    """PolyCommitment.toBytes: x || y big-endian canonical (src/zkvm/commitment_types.zig:49-54)."""
    if inf:
        return bytes(64)
    return fp_to_int(xy[:4]).to_bytes(32, "big") + fp_to_int(xy[4:]).to_bytes(32, "big")


__all__ = [_k for _k in dir() if not _k.startswith("__")]  # underscore helpers are shared between the parts too
