"""sumcheck.py — zolt.subprotocols: Sumcheck prover / verifier, sharded sumcheck, runSumcheck.

Part of the zolt_amd.api package (the host mirror of the reference's module API over libzolt_gpu.so); import zolt_amd.api,
which re-exports every name of every part."""
import numpy as np

from .. import lib
from ._base import *  # noqa: F401,F403
from .msm import *  # noqa: F401,F403
from .commitment import *  # noqa: F401,F403
from .wire import *  # noqa: F401,F403
from .poly import *  # noqa: F401,F403

# ---- sumcheck
class SumcheckVerificationFailed(Exception):
    pass


class Sumcheck:
    class Prover:
        """Sumcheck(F).Prover with the polynomial resident on the GPU (src/subprotocols/mod.zig:50-134)."""

        def __init__(self, polynomial):
            self._s = lib.SumcheckSession.open(polynomial.evaluations, lib.SC_HIGH_HALF)
            self.round = 0

        def nextRound(self):
            """-> coeffs [g(0), g(1) - g(0)] (src/subprotocols/mod.zig:69-109)."""
            g0, g1 = self._s.round_sums()
            c1 = _limbs((_int(g1) - _int(g0)) % R_MOD)  # Montgomery form is linear: sub on limbs mod r
            return np.stack([g0, c1])

        def receiveChallenge(self, challenge):
            self._s.bind(challenge)
            self.round += 1

        def isComplete(self):
            return len(self._s) == 1

        def getFinalEval(self):
            assert self.isComplete()
            return self._s.final()

        def deinit(self):
            self._s.close()

    class Verifier:
        """Toy verifier with the deterministic 64-bit mixer (src/subprotocols/mod.zig:137-244)."""

        def __init__(self, claim):
            self.claim = np.asarray(claim, dtype=np.uint64)
            self.round = 0
            self.challenges = []

        @staticmethod
        def _eval(coeffs, x_int):
            """UniPoly.evaluate by Horner (src/poly/mod.zig:608-618) on canonical ints."""
            res = 0
            for c in reversed(coeffs):
                res = (res * x_int + c) % R_MOD
            return res

        def deriveChallenge(self, coeffs):
            h = 0x9E3779B97F4A7C15
            h ^= self.round
            h = (h * 0xFF51AFD7ED558CCD) & _M64
            for limb in self.claim:
                h ^= int(limb)
                h = (h * 0xC4CEB9FE1A85EC53) & _M64
            for c in coeffs:
                for limb in c:
                    h ^= int(limb)
                    h = (h * 0xFF51AFD7ED558CCD) & _M64
                    h ^= h >> 33
            h ^= h >> 33
            h = (h * 0xFF51AFD7ED558CCD) & _M64
            h ^= h >> 33
            return h

        def verifyRound(self, coeffs):
            ci = [fr_to_int(c) for c in coeffs]
            p0, p1 = self._eval(ci, 0), self._eval(ci, 1)
            if (p0 + p1) % R_MOD != fr_to_int(self.claim):
                raise SumcheckVerificationFailed()
            h = self.deriveChallenge(coeffs)
            challenge = fr_from_int(h)
            self.challenges.append(challenge)
            self.claim = fr_from_int(self._eval(ci, h))
            self.round += 1
            return challenge


def sumcheck_shard_slice(length, world, rank, layout):
    """Index set of rank `rank` for a table sharded over `world` (a power of two) ranks such that every fold of
    the first log2(length/world) rounds is local:
      LOW_PAIR  (pairs 2i,2i+1 — DensePolynomial.bindLow, src/poly/mod.zig:160-175): contiguous chunk (high bits);
      HIGH_HALF (pairs i,i+half — bindFirst, :128-149): the residue class i ≡ rank (mod world) (low bits).
    Returns a slice usable on the full table."""
    assert world & (world - 1) == 0 and length % world == 0
    if layout == lib.SC_LOW_PAIR:
        c = length // world
        return slice(rank * c, (rank + 1) * c)
    return slice(rank, length, world)


def sharded_eq_args(r, world, rank, layout, scaling_factor=None):
    """Arguments (r_local, scale_local) with which a rank builds ITS shard of EqPolynomial.evals(r) (big-endian:
    r[0] <-> MSB, src/poly/mod.zig:252-290) with one ordinary eq-table build: the shard is the eq table of the
    remaining variables scaled by eq(shared variables, rank bits) — the "shared prefix scalar" of SURVEY §8(e).
    Host arithmetic on log2(world) challenges only."""
    g = world.bit_length() - 1
    v = len(r)
    assert g <= v
    ri = [fr_to_int(x) for x in r]
    # LOW_PAIR shards by the high index bits = r[0..g); HIGH_HALF shards by the low index bits = r[v-g..v)
    shared, rest = (ri[:g], r[g:]) if layout == lib.SC_LOW_PAIR else (ri[v - g:], r[:v - g])
    sc = fr_to_int(scaling_factor) if scaling_factor is not None else 1
    for j, rv in enumerate(shared):
        bit = (rank >> (g - 1 - j)) & 1
        sc = sc * (rv if bit else (1 - rv)) % R_MOD
    return np.asarray(rest, dtype=np.uint64).reshape(-1, 4), fr_from_int(sc)


class ShardedSumcheck:
    """Sumcheck(F).Prover (src/subprotocols/mod.zig:50-134) over a table sharded across GPUs.

    Every rank holds `sumcheck_shard_slice` of the table in an ordinary device session; a round is: local pair of
    sums -> ONE all-gather of 64 B per rank -> modular sum on the host (field addition is exact, so the order of the
    ranks does not matter) -> the caller's transcript derives the challenge -> local fold. After
    log2(len/world) rounds each rank is left with one element; those `world` elements are all-gathered once and
    the last log2(world) rounds run redundantly on every rank in a tiny session. Outputs are the reference's.

    `backend` supplies the device operations (GpuSumcheckShardBackend; the gloo CPU test plugs in a CPU-side one):
        round_sums() -> torch int64[8] (local g0||g1)      bind(challenge)      local_len()
        residual()   -> torch int64[4]                     open_tail(table u64[world,4]) -> session-like
    """

    def __init__(self, backend, world_size, rank, group=None):
        assert world_size & (world_size - 1) == 0, "world size must be a power of two"
        self.backend, self.world, self.rank, self.group = backend, world_size, rank, group
        self._tail = None
        self.round = 0
        if self.backend.local_len() == 1:
            self._enter_tail()

    def _all_gather(self, t):
        import torch
        import torch.distributed as dist
        if self.world == 1 and not dist.is_initialized():
            return t.reshape(1, -1)
        if dist.get_backend(self.group) == "gloo" and t.is_cuda:  # several ranks sharing one GPU (debugging / tests)
            parts = [torch.empty(t.numel(), dtype=torch.int64) for _ in range(self.world)]
            dist.all_gather(parts, t.cpu(), group=self.group)
            return torch.stack(parts)
        out = torch.empty((self.world, t.numel()), dtype=torch.int64, device=t.device)
        dist.all_gather_into_tensor(out, t.reshape(1, -1), group=self.group)
        return out

    def _enter_tail(self):
        res = self._all_gather(self.backend.residual()).cpu().numpy().view(np.uint64).reshape(self.world, 4)
        self._tail = self.backend.open_tail(res) if self.world > 1 else None
        self._final = res[0].copy() if self.world == 1 else None

    def nextRound(self):
        """-> coeffs [g(0), g(1) - g(0)] of the WHOLE table."""
        if self._tail is not None:
            g0, g1 = self._tail.round_sums()
            a, b = _int(g0), _int(g1)
        else:
            rec = self._all_gather(self.backend.round_sums()).cpu().numpy().view(np.uint64).reshape(self.world, 8)
            a = sum(_int(x[:4]) for x in rec) % R_MOD
            b = sum(_int(x[4:]) for x in rec) % R_MOD
        return np.stack([_limbs(a), _limbs((b - a) % R_MOD)])

    def receiveChallenge(self, challenge):
        if self._tail is not None:
            self._tail.bind(challenge)
        else:
            self.backend.bind(challenge)
            if self.backend.local_len() == 1:
                self._enter_tail()
        self.round += 1

    def isComplete(self):
        if self._tail is not None:
            return len(self._tail) == 1
        return self.backend.local_len() == 1 and self.world == 1

    def getFinalEval(self):
        assert self.isComplete()
        return self._tail.final() if self._tail is not None else self._final

    def deinit(self):
        if self._tail is not None:
            self._tail.close()
        self.backend.close()


class GpuSumcheckShardBackend:
    """ShardedSumcheck backend over libzolt_gpu.so. `d_local` is this rank's shard (torch CUDA int64[len,4], Montgomery
    limbs); the session runs on torch's CURRENT stream so that the RCCL all-gather (torch.distributed orders it
    against that stream) follows the sums kernel without a host synchronisation."""

    def __init__(self, d_local, layout):
        import torch
        self.layout = layout
        self.dev = d_local.device
        self._st = GpuShardBackend._stream()
        self._s = lib.SumcheckSession.open_dev(d_local.data_ptr(), d_local.shape[0], layout, stream=self._st)

    def local_len(self):
        return len(self._s)

    def round_sums(self):
        import torch
        out = torch.empty(8, dtype=torch.int64, device=self.dev)
        self._s.round_sums_dev(out.data_ptr())
        return out

    def bind(self, challenge):
        self._s.bind(challenge)

    def residual(self):
        import torch
        out = torch.empty(4, dtype=torch.int64, device=self.dev)
        self._s.read_dev(out.data_ptr())
        return out

    def open_tail(self, table):
        return lib.SumcheckSession.open(table, self.layout)

    def close(self):
        self._s.close()


def runSumcheck(polynomial):
    """runSumcheck (src/subprotocols/mod.zig:302-354) -> dict(claim, rounds, final_point, final_eval, result), with the
    prover AND the toy verifier on the device (zg_run_sumcheck): no PCIe crossing between rounds. Raises
    SumcheckVerificationFailed where the reference returns that error."""
    try:
        out = lib.run_sumcheck(polynomial.evaluations)
    except lib.SumcheckVerificationFailed as e:
        raise SumcheckVerificationFailed(str(e)) from None
    out["rounds"] = list(out["rounds"])
    out["final_point"] = list(out["final_point"])
    return out


def runSumcheckInteractive(polynomial):
    """The same protocol with the verifier on the host and one device round trip per round — the shape a prover with
    a real (Keccak/Blake2b) transcript has. Same outputs as runSumcheck."""
    s = lib.SumcheckSession.open(polynomial.evaluations, lib.SC_HIGH_HALF)
    g0, g1 = s.round_sums() if polynomial.num_vars else (polynomial.evaluations[0], np.zeros(4, dtype=np.uint64))
    s.close()
    claim = _limbs((_int(g0) + _int(g1)) % R_MOD)
    prover = Sumcheck.Prover(polynomial)
    verifier = Sumcheck.Verifier(claim)
    rounds = []
    for _ in range(polynomial.num_vars):
        coeffs = prover.nextRound()
        ch = verifier.verifyRound(coeffs)
        prover.receiveChallenge(ch)
        rounds.append(coeffs)
    final_eval = prover.getFinalEval()
    prover.deinit()
    return {"claim": claim, "rounds": rounds, "final_point": list(verifier.challenges), "final_eval": final_eval,
            "result": bool(np.array_equal(verifier.claim, final_eval))}


__all__ = [_k for _k in dir() if not _k.startswith("__")]  # underscore helpers are shared between the parts too
