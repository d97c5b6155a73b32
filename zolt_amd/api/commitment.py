"""commitment.py — zolt.poly.commitment: HyperKZG, Dory row commitments.

Part of the zolt_amd.api package (the host mirror of the reference's module API over libzolt_gpu.so); import zolt_amd.api,
which re-exports every name of every part."""
import numpy as np

from .. import lib
from ._base import *  # noqa: F401,F403
from .msm import *  # noqa: F401,F403

# ---- HyperKZG (commit side)
class HyperKZG:
    TAU = 0x12345678  # src/poly/commitment/mod.zig:189 (mock SRS, INSECURE by design)

    class SetupParams:
        def __init__(self, xy, inf, sharded=False, dev=None):
            self.powers_of_tau_g1 = xy
            self.infinity = inf
            self.max_degree = xy.shape[0]
            # device-resident for the whole run (:122-140); sharded=True: one shard per GPU bound by lib.init_devices — commit and
            # batchCommit then go through the one-process multi-GPU entry points (zg_msm_g1_sharded / zg_msm_g1_batch_sharded)
            self.sharded = bool(sharded)
            self._dev = dev if dev is not None else (lib.ShardedBases.upload(xy, inf) if sharded else lib.Bases.upload(xy, inf))

        def deinit(self):
            self._dev.free()

    @staticmethod
    def setup(max_degree, expected_uses=0):
        """powers[i] = scalarMul(G1, tau^i).toAffine() (src/poly/commitment/mod.zig:174-213). expected_uses 1..15: a key that serves one
        proof — no table of multiples (three commits and an opening are fewer MSMs than its break-even); 0: an SRS that lives on."""
        # the powers tau^i (:196-198), the fixed-base batch (every product has the same base: 32 table additions per point instead of
        # double-and-add) and the MSM handle with its table of multiples are built on the device (zg_hyperkzg_setup); the points come back
        # once, for SetupParams.powers_of_tau_g1
        dev, xy, inf = lib.Bases.hyperkzg_setup(generator(), fr_from_int(HyperKZG.TAU), max_degree, expected_uses=expected_uses)
        return HyperKZG.SetupParams(xy, inf, dev=dev)

    @staticmethod
    def commit(params, evals):
        """commit(params, evals) (src/poly/commitment/mod.zig:239-255): empty -> identity; n = min(len, srs)."""
        evals = np.ascontiguousarray(evals, dtype=np.uint64).reshape(-1, 4)
        if evals.shape[0] == 0:
            return np.zeros(8, dtype=np.uint64), 1
        n = min(evals.shape[0], params.max_degree)
        if params.sharded:
            return params._dev.msm(evals[:n], n=n)
        return params._dev.msm(evals[:n], off=0, n=n)

    @staticmethod
    def commitU64(params, values):
        """commit to the polynomial whose evaluations are F.fromU64 of `values` (machine words) — what commitBytecode / commitMemory /
        commitRegisters build (src/zkvm/mod.zig:1518-1617): the words cross as they are (zg_msm_g1_u64), same commitment bytes"""
        v = np.ascontiguousarray(values, dtype=np.uint64).reshape(-1)
        if v.size == 0:
            return np.zeros(8, dtype=np.uint64), 1
        n = min(v.size, params.max_degree)
        if params.sharded:  # the sharded handle takes field elements
            return HyperKZG.commit(params, lib.field_op(lib.FR, lib.OP_TO_MONT, np.stack([v[:n], np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.zeros(n, np.uint64)], axis=1)))
        return params._dev.msm_u64(v[:n], n=n)

    @staticmethod
    def batchCommit(params, polys):
        """batchCommit (src/poly/commitment/mod.zig:558-570): out[i] = commit(poly_i). Polynomials of equal (clamped)
        length share one zg_msm_g1_batch call, which fuses short vectors into a single launch set."""
        polys = [np.ascontiguousarray(p, dtype=np.uint64).reshape(-1, 4) for p in polys]
        out = [None] * len(polys)
        groups = {}
        for i, p in enumerate(polys):
            groups.setdefault(min(p.shape[0], params.max_degree), []).append(i)
        for n, idx in groups.items():
            if n == 0 or len(idx) == 1:
                for i in idx:
                    out[i] = HyperKZG.commit(params, polys[i])
            else:
                xy, inf = params._dev.msm_batch([polys[i][:n] for i in idx], n=n)
                for j, i in enumerate(idx):
                    out[i] = (xy[j], int(inf[j]))
        return out

    @staticmethod
    def open(params, evals, point, value):
        """open (src/poly/commitment/mod.zig:261-324): per variable commit(q = hi - lo), fold high half.
        Returns (quotient commitments [(xy, inf)], final_eval)."""
        point = np.ascontiguousarray(point, dtype=np.uint64).reshape(-1, 4)
        if point.shape[0] == 0:
            return [], np.asarray(value, dtype=np.uint64)
        q, qinf, final = lib.hyperkzg_open(params._dev, evals, point, value)  # whole loop resident on the device
        return [(q[i], int(qinf[i])) for i in range(point.shape[0])], final


    @staticmethod
    def batchOpen(params, polys, point):
        """batchOpen (src/poly/commitment/mod.zig:607-732) -> dict(quotient_commitments [(xy, inf)], evaluations,
        final_eval, batching_challenge); the combination, the evaluations and the fold/commit loop run on the device."""
        point = np.ascontiguousarray(point, dtype=np.uint64).reshape(-1, 4)
        q, qinf, ev, fin, gam = lib.hyperkzg_batch_open(params._dev, polys, point)
        return {"quotient_commitments": [(q[i], int(qinf[i])) for i in range(q.shape[0])], "evaluations": ev, "final_eval": fin,
                "batching_challenge": gam}


class Dory:
    """The data-parallel G1 / Fr pieces of Dory's commit and open (src/poly/commitment/dory.zig; pairings and GT arithmetic stay the
    reference's): the row commitments are a batch of MSMs over one prefix of g1_vec, the vector-matrix product a weighted column sum."""

    @staticmethod
    def computeRowCommitments(g1_bases, evals, num_columns):
        """computeRowCommitments (:646-670): g1_bases = a lib.Bases handle over params.g1_vec (resident, like the HyperKZG SRS);
        row r = MSM(g1_vec[0..len(row)], row r of evals). Full rows go through ONE fused launch set (zg_msm_g1_batch), a shorter last
        row is one more MSM over the prefix -> (xy (rows, 8), inf (rows,))"""
        ev = np.ascontiguousarray(evals, dtype=np.uint64).reshape(-1, 4)
        full, rest = divmod(ev.shape[0], num_columns)
        assert num_columns <= g1_bases.n
        out = np.zeros((full + (1 if rest else 0), 8), dtype=np.uint64)
        inf = np.zeros(out.shape[0], dtype=np.uint8)
        if full:
            out[:full], inf[:full] = g1_bases.msm_batch([ev[r * num_columns:(r + 1) * num_columns] for r in range(full)], n=num_columns)
        if rest:
            xy, i = g1_bases.msm(ev[full * num_columns:], n=rest)
            out[full], inf[full] = xy, i
        return out, inf

    @staticmethod
    def multilinearLagrangeBasis(point, out_len=None):
        """multilinearLagrangeBasis (:544-588): the eq table of the point with the index's LOW bit on point[0] — the device's eq table of
        the reversed point; a shorter output is its first entries -> (out_len, 4)"""
        pt = np.ascontiguousarray(point, dtype=np.uint64).reshape(-1, 4)
        full = lib.fr_eq_table(np.ascontiguousarray(pt[::-1])) if pt.shape[0] else fr_from_int(1).reshape(1, 4)
        return full if out_len is None else np.ascontiguousarray(full[:out_len])

    @staticmethod
    def computeEvaluationVectors(point, nu, sigma):
        """computeEvaluationVectors (:590-620) -> (left_vec (2^nu, 4), right_vec (2^sigma, 4))"""
        pt = np.ascontiguousarray(point, dtype=np.uint64).reshape(-1, 4)
        d = pt.shape[0]
        left, right = np.zeros((1 << nu, 4), dtype=np.uint64), np.zeros((1 << sigma, 4), dtype=np.uint64)
        if d <= sigma:
            right[:1 << d] = Dory.multilinearLagrangeBasis(pt)
            left[0] = fr_from_int(1)
        elif d <= nu + sigma:
            right[:] = Dory.multilinearLagrangeBasis(pt[:sigma])
            left[:1 << (d - sigma)] = Dory.multilinearLagrangeBasis(pt[sigma:])
        else:  # more variables than the matrix has: the row basis is cut at 2^nu entries
            right[:] = Dory.multilinearLagrangeBasis(pt[:sigma])
            left[:] = Dory.multilinearLagrangeBasis(pt[sigma:], 1 << nu)
        return left, right

    @staticmethod
    def computeVectorMatrixProduct(evals, left_vec, nu, sigma):
        """computeVectorMatrixProduct (:622-642): v = L^T M over the 2^nu x 2^sigma matrix of evaluations (zg_fr_weighted_colsum); rows
        past left_vec and entries past evals are zero -> (2^sigma, 4)"""
        rows, cols = 1 << nu, 1 << sigma
        ev = np.ascontiguousarray(evals, dtype=np.uint64).reshape(-1, 4)
        lv = np.ascontiguousarray(left_vec, dtype=np.uint64).reshape(-1, 4)
        m = np.zeros((rows * cols, 4), dtype=np.uint64)
        m[:min(ev.shape[0], rows * cols)] = ev[:rows * cols]
        w = np.zeros((rows, 4), dtype=np.uint64)
        w[:min(lv.shape[0], rows)] = lv[:rows]
        return lib.fr_weighted_colsum(m, rows, cols, w.reshape(1, rows, 4))[0]


__all__ = [_k for _k in dir() if not _k.startswith("__")]  # underscore helpers are shared between the parts too
