"""blake2b.py — Blake2b transcript of the original Stage-4 prover.

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
from .transcript import *  # noqa: F401,F403
from .provers import *  # noqa: F401,F403

class Blake2bTranscript:
    """Blake2bTranscript(F) — the Jolt-compatible transcript the reference's proving path uses (src/transcripts/blake2b.zig:25-545):
    a 32-byte running state and a round counter; every operation hashes state || [0u8; 28] || n_rounds_be32 || payload with
    Blake2b-256 and the digest becomes the new state. Host code (hashlib), like the Keccak one above."""

    def __init__(self, label=b"Jolt"):
        import hashlib
        label = bytes(label)
        assert len(label) < 33
        self._blake = lambda data: hashlib.blake2b(data, digest_size=32).digest()
        self.state = self._blake(label.ljust(32, b"\0"))  # :39-69
        self.n_rounds = 0

    def _hash_with(self, payload):
        """hasher() (:76-87) + payload, then updateState (:90-93)"""
        out = self._blake(self.state + bytes(28) + self.n_rounds.to_bytes(4, "big") + bytes(payload))
        self.state = out
        self.n_rounds += 1
        return out

    def appendMessage(self, msg):  # :96-120: right-padded to 32 bytes
        msg = bytes(msg)
        assert len(msg) < 33
        self._hash_with(msg.ljust(32, b"\0"))

    def appendBytes(self, data):  # :123-156
        self._hash_with(bytes(data))

    def appendU64(self, x):  # :160-176: [0u8; 24] ++ x.to_be_bytes()
        self._hash_with(bytes(24) + int(x).to_bytes(8, "big"))

    def appendScalar(self, scalar):  # :182-200: canonical value, little-endian bytes reversed to big-endian
        self.appendBytes(fr_to_int(scalar).to_bytes(32, "big"))

    def appendScalars(self, scalars):  # :205-211
        self.appendMessage(b"begin_append_vector")
        for s in np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4):
            self.appendScalar(s)
        self.appendMessage(b"end_append_vector")

    def challengeBytes(self, n):  # :215-240
        out = b""
        while n - len(out) > 32:
            out += self._hash_with(b"")
        return out + self._hash_with(b"")[:n - len(out)]

    def challengeU128(self):  # :243-254
        return int.from_bytes(self.challengeBytes(16)[::-1], "big")

    def challengeScalarFull(self):  # :279-312: the 128-bit value as a proper Montgomery element
        buf = self.challengeBytes(16)[::-1]
        return fr_from_int(int.from_bytes(buf, "little"))

    def challengeScalar(self):  # :264-266,332-390: 125-bit mask, stored as RAW limbs [0, 0, low, high] (MontU128Challenge)
        v = int.from_bytes(self.challengeBytes(16)[::-1], "big") & ((1 << 125) - 1)
        return np.array([0, 0, v & _M64, v >> 64], dtype=np.uint64)

    def challengeVector(self, n):  # :392-399 (challengeScalar each)
        return np.stack([self.challengeScalar() for _ in range(n)]) if n else np.zeros((0, 4), dtype=np.uint64)


__all__ = [_k for _k in dir() if not _k.startswith("__")]  # underscore helpers are shared between the parts too
