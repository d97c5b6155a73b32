"""msm.py — zolt.msm: MSM, BatchMSM, ParallelMSM, sharded orchestration.

Part of the zolt_amd.api package (the host mirror of the reference's module API over libzolt_gpu.so); import zolt_amd.api,
which re-exports every name of every part."""
import numpy as np

from .. import lib
from ._base import *  # noqa: F401,F403

# ---- MSM
class MSM:
    """MSM(F, G) with F = Fr, G = Fp."""

    @staticmethod
    def compute(bases_xy, scalars, bases_inf=None):
        """MSM.compute(bases, scalars) -> (xy, inf)   (src/msm/mod.zig:355-372)."""
        bases_xy = np.ascontiguousarray(bases_xy, dtype=np.uint64).reshape(-1, 8)
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        assert bases_xy.shape[0] == scalars.shape[0]  # std.debug.assert(bases.len == scalars.len), :359
        b = lib.Bases.upload(bases_xy, bases_inf, expected_uses=1)  # a one-shot slice: no precompute table
        try:
            return b.msm(scalars)
        finally:
            b.free()

    @staticmethod
    def scalarMul(base_xy, scalar, base_inf=0):
        """MSM.scalarMul(base, scalar).toAffine() (src/msm/mod.zig:503-540)."""
        out, inf = lib.g1_scalar_mul_batch(np.asarray(base_xy).reshape(1, 8), np.array([base_inf], dtype=np.uint8),
                                           np.asarray(scalar).reshape(1, 4))
        return out[0], int(inf[0])


class AffinePoint:
    """AffinePoint(G) group law (src/msm/mod.zig:15-140) on (xy[8], inf) pairs."""

    @staticmethod
    def add(a_xy, a_inf, b_xy, b_inf):
        """AffinePoint.add (:74-103) -> (xy, inf)"""
        out, inf = lib.g1_affine_add_batch(np.asarray(a_xy).reshape(1, 8), np.array([a_inf], dtype=np.uint8),
                                           np.asarray(b_xy).reshape(1, 8), np.array([b_inf], dtype=np.uint8))
        return out[0], int(inf[0])

    @staticmethod
    def double(xy, inf=0):
        """AffinePoint.double (:118-138) = add(p, p)"""
        return AffinePoint.add(xy, inf, xy, inf)

    @staticmethod
    def isOnCurve(xy, inf=0):
        return bool(lib.g1_is_on_curve_batch(np.asarray(xy).reshape(1, 8), np.array([inf], dtype=np.uint8))[0])


class ParallelMSM:
    """ParallelMSM.compute (src/msm/mod.zig:588-653) in the reference's process model: ONE process, its workers = the GPUs bound
    by lib.init_devices (contiguous chunks of ceil(n / S), one partial per GPU, RCCL all-gather, serial combine on device 0)."""

    @staticmethod
    def compute(bases_xy, scalars, bases_inf=None, num_threads=None):
        bases_xy = np.ascontiguousarray(bases_xy, dtype=np.uint64).reshape(-1, 8)
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        assert bases_xy.shape[0] == scalars.shape[0]
        sb = lib.ShardedBases.upload(bases_xy, bases_inf, precompute_levels=1)
        try:
            return sb.msm(scalars)
        finally:
            sb.free()


class ParallelBatchMSM:
    """ParallelBatchMSM.compute (src/msm/mod.zig:683-748) / HyperKZG.batchCommit sharded: k partials per GPU, one exchange."""

    @staticmethod
    def compute(bases_xy, scalar_batches, bases_inf=None):
        sb = lib.ShardedBases.upload(bases_xy, bases_inf, precompute_levels=1)
        try:
            return sb.msm_batch(scalar_batches)
        finally:
            sb.free()


class BatchMSM:
    @staticmethod
    def compute(bases_xy, scalar_batches, bases_inf=None):
        """BatchMSM.compute / ParallelBatchMSM.compute (src/msm/mod.zig:545-565,683-748)."""
        b = lib.Bases.upload(bases_xy, bases_inf)
        try:
            return b.msm_batch(scalar_batches)
        finally:
            b.free()


def shard_bounds(n, parts):
    """ParallelMSM's partition: contiguous chunks of ceil(n/T) (src/msm/mod.zig:609,619-639).
    Returns [(start, end)] of length `parts`; trailing shards may be empty."""
    chunk = (n + parts - 1) // parts if parts else 0
    out = []
    for i in range(parts):
        s = min(i * chunk, n)
        out.append((s, min(s + chunk, n)))
    return out


class ShardedMSM:
    """ParallelMSM across GPUs: rank r owns bases/scalars [start_r, end_r), computes its
    Jacobian partial on its GPU, the partials are all-gathered (RCCL via torch.distributed: one
    96-byte record per rank), and the serial combine + toAffine runs on the device.

    `backend` supplies the two device operations so the orchestration can be exercised on CPU
    with gloo in the tests:
        backend.partial(rank_scalars_tensor) -> torch int64[12] tensor (device of the backend)
        backend.combine(gathered int64[world,12]) -> (xy, inf)
    """

    def __init__(self, backend, world_size, rank, group=None):
        self.backend, self.world, self.rank, self.group = backend, world_size, rank, group

    def compute(self, local_scalars, out=None):
        """-> (xy, inf) on the host, or, with `out` (a device int64[9] slot: xy[8] + flag word), fully
        asynchronous: partial MSM, all-gather and combine are only stream-ordered."""
        import torch
        import torch.distributed as dist
        part = self.backend.partial(local_scalars)
        if self.world == 1 and not dist.is_initialized():
            gathered = part.reshape(1, 12)
        elif dist.get_backend(self.group) == "gloo" and part.is_cuda:
            # debugging aid (several ranks on one GPU): stage the 96-byte records through the host
            parts = [torch.empty(12, dtype=torch.int64) for _ in range(self.world)]
            dist.all_gather(parts, part.cpu(), group=self.group)
            gathered = torch.stack(parts).to(part.device)
        else:
            gathered = torch.empty((self.world, 12), dtype=torch.int64, device=part.device)
            dist.all_gather_into_tensor(gathered, part.reshape(1, 12), group=self.group)
        if out is not None:
            return self.backend.combine_async(gathered, out)
        return self.backend.combine(gathered)

    def compute_batch(self, local_scalar_sets, out=None):
        """m MSMs over the same bases behind ONE exchange (ParallelBatchMSM, src/msm/mod.zig:683-748): this rank's m partials go into
        one [m, 12] block, one all-gather of m * 96 bytes per rank replaces m collectives of 96 bytes (at 2^17 points per rank a
        partial takes 0.23 ms: 32 MSMs per step were 32 tiny collectives serialised on the communicator's stream), one launch combines
        the m results. -> list of (xy, inf), or with `out` (device int64[m, 9]) fully stream-ordered."""
        import torch
        import torch.distributed as dist
        m = len(local_scalar_sets)
        parts = self.backend.partial_batch(local_scalar_sets)  # [m, 12]
        if self.world == 1 and not dist.is_initialized():
            gathered = parts.reshape(1, m, 12)
        elif dist.get_backend(self.group) == "gloo" and parts.is_cuda:
            lst = [torch.empty((m, 12), dtype=torch.int64) for _ in range(self.world)]
            dist.all_gather(lst, parts.cpu(), group=self.group)
            gathered = torch.stack(lst).to(parts.device)
        else:
            gathered = torch.empty((self.world, m, 12), dtype=torch.int64, device=parts.device)
            dist.all_gather_into_tensor(gathered, parts.reshape(1, m, 12), group=self.group)
        if out is not None:
            return self.backend.combine_batch_async(gathered, out)
        return self.backend.combine_batch(gathered)


class GpuShardBackend:
    """ShardedMSM backend over libzolt_gpu.so; tensors are torch CUDA tensors (device memory plumbing). All work is
    enqueued on torch's CURRENT stream, so the caller can rotate streams (`with torch.cuda.stream(s)`) to overlap
    consecutive sharded MSMs; the collective is ordered against that stream by torch.distributed."""

    def __init__(self, bases, n_local):
        self.bases, self.n = bases, n_local

    @staticmethod
    def _stream():
        import torch
        s = torch.cuda.current_stream().cuda_stream
        assert s != 0, "run under an explicit torch stream: a NULL stream means the library's own stream"
        return s

    def partial(self, d_scalars):
        import torch
        out = torch.empty(12, dtype=torch.int64, device=d_scalars.device)
        # un-normalised Jacobian partial: the combine result is identical and the rank skips an inversion
        self.bases.msm_partial_fast_dev(d_scalars.data_ptr(), self.n, out.data_ptr(), stream=self._stream())
        return out

    def partial_batch(self, d_scalar_sets):
        import torch
        out = torch.empty((len(d_scalar_sets), 12), dtype=torch.int64, device=d_scalar_sets[0].device)
        for j, sc in enumerate(d_scalar_sets):
            self.bases.msm_partial_fast_dev(sc.data_ptr(), self.n, out[j].data_ptr(), stream=self._stream())
        return out

    def combine(self, gathered):
        return lib.combine_partials_dev(gathered.data_ptr(), gathered.shape[0], stream=self._stream())

    def combine_batch_async(self, gathered, out):
        """gathered: device int64[world, m, 12]; out: device int64[m, 9]"""
        gathered.record_stream(__import__("torch").cuda.current_stream())
        world, m = gathered.shape[0], gathered.shape[1]
        lib.combine_partials_batch_dev_async(gathered.data_ptr(), world, 12 * m, m, out.data_ptr(), stream=self._stream())
        return None

    def combine_batch(self, gathered):
        import torch
        m = gathered.shape[1]
        out = torch.empty((m, 9), dtype=torch.int64, device=gathered.device)
        self.combine_batch_async(gathered, out)
        torch.cuda.current_stream().synchronize()
        h = out.cpu().numpy().view(np.uint64)
        return [(h[j, :8].copy(), int(h[j, 8] & 0xFF)) for j in range(m)]

    def combine_async(self, gathered, out):
        gathered.record_stream(__import__("torch").cuda.current_stream())
        lib.combine_partials_dev_async(gathered.data_ptr(), gathered.shape[0], out.data_ptr(), out[8:].data_ptr(), stream=self._stream())
        return None


__all__ = [_k for _k in dir() if not _k.startswith("__")]  # underscore helpers are shared between the parts too
