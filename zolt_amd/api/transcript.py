"""transcript.py — Keccak Fiat-Shamir transcript (host, between rounds).

Part of the zolt_amd.api package (the host mirror of the reference's module API over libzolt_gpu.so); import zolt_amd.api,
which re-exports every name of every part."""
import numpy as np

from .. import lib
from ._base import *  # noqa: F401,F403
from .msm import *  # noqa: F401,F403
from .commitment import *  # noqa: F401,F403
from .wire import *  # noqa: F401,F403
from .poly import *  # noqa: F401,F403
from .sumcheck import *  # noqa: F401,F403

# ---- host Fiat-Shamir transcript + the prover fold sites driven by it
_KECCAK_RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808a, 0x8000000080008000, 0x000000000000808b, 0x0000000080000001,
              0x8000000080008081, 0x8000000000008009, 0x000000000000008a, 0x0000000000000088, 0x0000000080008009, 0x000000008000000a,
              0x000000008000808b, 0x800000000000008b, 0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080,
              0x000000000000800a, 0x800000008000000a, 0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
_KECCAK_ROTC = [1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44]
_KECCAK_PILN = [10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1]


class Transcript:
    """Transcript(F) — the reference's Keccak Fiat-Shamir transcript (src/transcripts/mod.zig:49-221), on the host, where it stays
    in the reference integration too: a sequential hash between rounds. Field elements go in as their raw Montgomery limbs
    (appendScalar, :100-110) and challenges come out through F.fromBytes (:116-130), so it plugs straight onto the C ABI's limbs."""
    KECCAK_RATE = 136

    def __init__(self, domain=b"Jolt"):
        self.state = bytearray(200)
        self.position = 0
        self.appendBytes(domain)

    def _keccakF(self):
        st = [int.from_bytes(self.state[8 * i:8 * i + 8], "little") for i in range(25)]
        for rc in _KECCAK_RC:
            bc = [st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20] for i in range(5)]
            for i in range(5):
                b1 = bc[(i + 1) % 5]
                t = bc[(i + 4) % 5] ^ (((b1 << 1) | (b1 >> 63)) & _M64)
                for j in range(i, 25, 5):
                    st[j] ^= t
            t = st[1]
            for i in range(24):
                j, n = _KECCAK_PILN[i], _KECCAK_ROTC[i]
                st[j], t = ((t << n) | (t >> (64 - n))) & _M64, st[j]
            for row in range(0, 25, 5):
                b = st[row:row + 5]
                for i in range(5):
                    st[row + i] = b[i] ^ ((~b[(i + 1) % 5]) & _M64 & b[(i + 2) % 5])
            st[0] ^= rc
        for i, v in enumerate(st):
            self.state[8 * i:8 * i + 8] = v.to_bytes(8, "little")

    def appendBytes(self, data):
        for byte in bytes(data):
            self.state[self.position] ^= byte
            self.position += 1
            if self.position >= self.KECCAK_RATE:
                self._keccakF()
                self.position = 0

    def appendMessage(self, label, message):
        self.appendBytes(label)
        self.appendBytes(message)

    def appendScalar(self, label, scalar):
        self.appendBytes(label)
        self.appendBytes(np.ascontiguousarray(scalar, dtype="<u8").tobytes())

    def appendScalars(self, label, scalars):
        self.appendBytes(label)
        for s in np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4):
            self.appendScalar(b"", s)

    def challengeScalar(self, label):
        self.appendBytes(label)
        self._keccakF()
        raw = np.frombuffer(bytes(self.state[:32]), dtype="<u8").astype(np.uint64)
        return _limbs((_int(raw) % R_MOD) * _MONT_R % R_MOD)  # F.fromBytes: the 256-bit little-endian integer, reduced, Montgomery

    def challengeScalars(self, label, count):
        self.appendBytes(label)
        return np.stack([self.challengeScalar(b"") for _ in range(count)]) if count else np.zeros((0, 4), dtype=np.uint64)

    def challengeBytes(self, label, n):
        """challengeBytes (:143-160): one Keccak-f per 136 output bytes, each block read from the start of the state"""
        self.appendBytes(label)
        out = b""
        while len(out) < n:
            self._keccakF()
            out += bytes(self.state[:min(n - len(out), self.KECCAK_RATE)])
        return out


def _fr_add(a, b):
    return _limbs((_int(a) + _int(b)) % R_MOD)


def _fr_sub(a, b):
    return _limbs((_int(a) - _int(b)) % R_MOD)


__all__ = [_k for _k in dir() if not _k.startswith("__")]  # underscore helpers are shared between the parts too
