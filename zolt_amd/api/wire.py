"""wire.py — ZOLT proof container and SRS wire formats.

Part of the zolt_amd.api package (the host mirror of the reference's module API over libzolt_gpu.so); import zolt_amd.api,
which re-exports every name of every part."""
import numpy as np

from .. import lib
from ._base import *  # noqa: F401,F403
from .msm import *  # noqa: F401,F403
from .commitment import *  # noqa: F401,F403

# ---- ZOLT v1 proof container: the commitments this backend produces
def parse_zolt_proof_commitments(data):
    """Header of serializeProof (src/zkvm/serialization.zig:283-306): "ZOLT" | u32 version | bytecode proof
    {commitment, read_ts, write_ts (64 B each, x||y big-endian, identity = 64 zero bytes,
    src/zkvm/commitment_types.zig:49-54), legacy field element 32 B} | memory proof {4 x 64 B} | register proof
    {4 x 64 B}. Returns {name: 64 raw bytes}; the rest of the proof (R1CS / stage proofs) is not parsed."""
    if len(data) < 8 + 3 * 64 + 32 + 8 * 64 or data[:4] != b"ZOLT":
        raise ValueError("not a ZOLT proof")
    version = int.from_bytes(data[4:8], "little")
    if version != 1:
        raise ValueError(f"unsupported ZOLT proof version {version}")
    out, off = {}, 8
    for name in ("bytecode.commitment", "bytecode.read_ts_commitment", "bytecode.write_ts_commitment"):
        out[name] = bytes(data[off:off + 64])
        off += 64
    off += 32  # bytecode._legacy_commitment
    for group in ("memory", "register"):
        for name in ("commitment", "final_state_commitment", "read_ts_commitment", "write_ts_commitment"):
            out[f"{group}.{name}"] = bytes(data[off:off + 64])
            off += 64
    return out


def serialize_zolt_proof_header(commitments):
    """The part of serializeProof (src/zkvm/serialization.zig:283-306) this backend produces: "ZOLT" | u32 version 1 | bytecode
    proof {commitment, read_ts, write_ts, 32-byte legacy field element} | memory proof {commitment, final_state, read_ts, write_ts}
    | register proof {same four}. `commitments`: {name: (xy, inf)} with the names of parse_zolt_proof_commitments; missing names are
    the identity (64 zero bytes, PolyCommitment.zero()) — what commitBytecode / commitMemory / commitRegisters leave in the
    timestamp and final-state slots (src/zkvm/mod.zig:1540-1546,1574-1581,1609-1616). Returns the first 744 bytes of the proof."""
    def enc(name):
        c = commitments.get(name)
        return bytes(64) if c is None else commitment_to_bytes(c[0], c[1])
    out = b"ZOLT" + (1).to_bytes(4, "little")
    for name in ("bytecode.commitment", "bytecode.read_ts_commitment", "bytecode.write_ts_commitment"):
        out += enc(name)
    out += bytes(32)  # bytecode._legacy_commitment = F.zero()
    for group in ("memory", "register"):
        for name in ("commitment", "final_state_commitment", "read_ts_commitment", "write_ts_commitment"):
            out += enc(f"{group}.{name}")
    return out


# ---- SRS wire format (G1 section)
class SRSError(Exception):
    pass


def srs_g1_from_raw(data):
    """G1 part of loadFromRawBinary (src/poly/commitment/srs.zig:256-306): u32 n (LE) | n x (x BE 32 B | y BE 32 B).
    All-zero 64 bytes = point at infinity (parseG1Uncompressed, :65-99). Coordinates are reduced and converted to
    Montgomery form on the GPU (Fp.fromBytesBE -> fromBytes, src/field/mod.zig:171-210). The G2 / generator trailer
    (pairing side, out of scope) is returned untouched. -> (xy (n,8) uint64, inf (n,) uint8, trailer bytes)"""
    if len(data) < 4:
        raise SRSError("TruncatedData")
    n = int.from_bytes(data[:4], "little")
    if len(data) < 4 + 64 * n + 128 + 64 + 128:
        raise SRSError("TruncatedData")
    body = np.frombuffer(data, dtype=np.uint8, count=64 * n, offset=4).reshape(n, 2, 32)
    inf = (~body.reshape(n, 64).any(axis=1)).astype(np.uint8)
    raw = np.ascontiguousarray(body[:, :, ::-1]).view(np.uint64).reshape(n, 8)  # BE bytes -> LE limbs
    xy = lib.field_op(lib.FP, lib.OP_TO_MONT, raw.reshape(2 * n, 4)).reshape(n, 8) if n else np.zeros((0, 8), dtype=np.uint64)
    xy[inf == 1] = 0
    if n and not lib.g1_is_on_curve_batch(xy, inf).all():  # parseG1Uncompressed, :93-96
        raise SRSError("PointNotOnCurve")
    return xy, inf, bytes(data[4 + 64 * n:])


PTAU_MAGIC = b"ptau"
_PTAU_HEADER, _PTAU_TAU_G1, _PTAU_TAU_G2, _PTAU_ALPHA_G1, _PTAU_BETA_G1, _PTAU_BETA_G2 = 1, 2, 3, 4, 5, 6


def _g1_from_le(sec, count):
    """parseG1LE (src/poly/commitment/srs.zig:616-660) over `count` 64-byte records: x | y as little-endian integers
    (reduced like Fp.fromBytesBE of the reversed bytes), all-zero = infinity, every other point checked on the curve.
    Conversion to Montgomery form and the curve check run on the GPU."""
    body = np.frombuffer(sec, dtype=np.uint8, count=64 * count).reshape(count, 64)
    inf = (~body.any(axis=1)).astype(np.uint8)
    raw = np.ascontiguousarray(body).view(np.uint64).reshape(count, 8)  # LE bytes are already LE limbs
    xy = lib.field_op(lib.FP, lib.OP_TO_MONT, raw.reshape(2 * count, 4)).reshape(count, 8) if count else np.zeros((0, 8), dtype=np.uint64)
    xy[inf == 1] = 0
    if count and not lib.g1_is_on_curve_batch(xy, inf).all():
        raise SRSError("PointNotOnCurve")
    return xy, inf


def srs_g1_from_ptau(data):
    """G1 side of loadFromPtau (src/poly/commitment/srs.zig:733-900, snarkjs powers-of-tau container): "ptau" | u32 version
    (= 1) | u32 sections | sections (u32 type, u64 size, payload). Header payload: u32 field size (= 32) | 32-byte prime |
    u32 power | u32 ceremony power. TauG1 holds min(2*2^power - 1, size/64) points, AlphaTauG1 / BetaTauG1 min(2^power,
    size/64). The G2 sections (pairing side, out of scope) are returned as raw bytes.
    -> dict(power, ceremony_power, powers_of_tau_g1=(xy, inf), alpha_tau_g1, beta_tau_g1 (or None), tau_g2_raw, beta_g2_raw)"""
    if len(data) < 12:
        raise SRSError("TruncatedData")
    if data[:4] != PTAU_MAGIC:
        raise SRSError("InvalidFileFormat")
    if int.from_bytes(data[4:8], "little") != 1:
        raise SRSError("UnsupportedFormat")
    nsec = int.from_bytes(data[8:12], "little")
    off, secs = 12, {}
    for _ in range(nsec):
        if off + 12 > len(data):
            raise SRSError("TruncatedData")
        typ = int.from_bytes(data[off:off + 4], "little")
        size = int.from_bytes(data[off + 4:off + 12], "little")
        off += 12
        if off + size > len(data):
            raise SRSError("TruncatedData")
        secs[typ] = data[off:off + size]  # a later section of the same type wins, as in the reference's scan
        off += size
    if _PTAU_HEADER not in secs:
        raise SRSError("InvalidFileFormat")
    hdr = secs[_PTAU_HEADER]
    if len(hdr) < 8:
        raise SRSError("TruncatedData")
    if int.from_bytes(hdr[:4], "little") != 32:
        raise SRSError("UnsupportedFormat")
    if len(hdr) < 44:
        raise SRSError("TruncatedData")
    power = int.from_bytes(hdr[36:40], "little")
    out = {"power": power, "ceremony_power": int.from_bytes(hdr[40:44], "little"),
           "powers_of_tau_g1": (np.zeros((0, 8), dtype=np.uint64), np.zeros(0, dtype=np.uint8)),
           "alpha_tau_g1": None, "beta_tau_g1": None,
           "tau_g2_raw": bytes(secs.get(_PTAU_TAU_G2, b"")), "beta_g2_raw": bytes(secs.get(_PTAU_BETA_G2, b""))}
    if _PTAU_TAU_G1 in secs:
        sec = secs[_PTAU_TAU_G1]
        out["powers_of_tau_g1"] = _g1_from_le(sec, min((1 << power) * 2 - 1, len(sec) // 64))
    for key, typ in (("alpha_tau_g1", _PTAU_ALPHA_G1), ("beta_tau_g1", _PTAU_BETA_G1)):
        if typ in secs:
            out[key] = _g1_from_le(secs[typ], min(1 << power, len(secs[typ]) // 64))
    return out


def srs_g1_to_raw(xy, inf, trailer=bytes(128 + 64 + 128)):
    """serializeToRawBinary's G1 section (src/poly/commitment/srs.zig:358-408): toBytesBE of x and y."""
    xy = np.ascontiguousarray(xy, dtype=np.uint64).reshape(-1, 8)
    n = xy.shape[0]
    canon = lib.field_op(lib.FP, lib.OP_FROM_MONT, xy.reshape(2 * n, 4)) if n else np.zeros((0, 4), dtype=np.uint64)
    be = np.ascontiguousarray(canon).view(np.uint8).reshape(n, 2, 32)[:, :, ::-1]
    return n.to_bytes(4, "little") + np.ascontiguousarray(be).tobytes() + bytes(trailer)


__all__ = [_k for _k in dir() if not _k.startswith("__")]  # underscore helpers are shared between the parts too
