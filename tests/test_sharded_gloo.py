"""world_size-2 gloo test of the multi-GPU orchestration (zolt_amd.api.ShardedMSM) on CPU.

The sharding, the all-gather of 96-byte Jacobian partials and the combine order are the product's;
the two device operations are supplied here by a test backend built on the CPU oracle so the
N > 1 control path runs without GPUs (on the GPU box the same class runs over RCCL with
GpuShardBackend — bench.py --gpus N)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleShardBackend:
    def __init__(self, xy):
        from oracle import binding as ob
        self.ob, self.xy = ob, xy

    def partial(self, scalars_t):
        ob = self.ob
        sc = scalars_t.numpy().view(np.uint64).reshape(-1, 4)
        out, inf = ob.msm_g1(self.xy, None, sc)
        one = ob.f_from_u64(ob.FP, np.array([1], dtype=np.uint64))[0]
        rec = np.concatenate([one, one, np.zeros(4, dtype=np.uint64)]) if inf else np.concatenate([out, one])
        return torch.from_numpy(rec.view(np.int64).copy())

    def combine(self, gathered):
        ob = self.ob
        recs = gathered.numpy().view(np.uint64).reshape(-1, 12)
        acc = np.concatenate([ob.f_from_u64(ob.FP, np.array([1, 1], dtype=np.uint64)).reshape(-1), np.zeros(4, dtype=np.uint64)])
        for r in recs:
            acc = ob.g1_jac_add(acc, r)
        return ob.g1_jac_to_affine(acc)

    def partial_batch(self, scalar_sets):
        return torch.stack([self.partial(sc) for sc in scalar_sets])

    def combine_batch(self, gathered):  # [world, m, 12]
        return [self.combine(gathered[:, j, :].contiguous()) for j in range(gathered.shape[1])]


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import binding as ob
    from tests import util as U
    from zolt_amd import api
    gm = ob.g1_gen_multiples(n)
    sc = ob.f_to_mont(ob.FR, U.random_raw256(4242, n))
    s, e = api.shard_bounds(n, world)[rank]
    sharded = api.ShardedMSM(OracleShardBackend(gm[s:e]), world, rank)
    xy, inf = sharded.compute(torch.from_numpy(sc[s:e].view(np.int64).copy()))
    want, winf = ob.msm_g1_parallel(gm, None, sc, world)   # the reference's ParallelMSM with T = world
    full, finf = ob.msm_g1(gm, None, sc)
    ok = inf == winf == finf and np.array_equal(xy, want) and np.array_equal(xy, full)
    # the batched form: m scalar vectors, ONE all-gather of m * 96 bytes per rank, m combines
    vecs = [ob.f_to_mont(ob.FR, U.random_raw256(4300 + j, n)) for j in range(3)]
    res = sharded.compute_batch([torch.from_numpy(v[s:e].view(np.int64).copy()) for v in vecs])
    for v, (bxy, binf) in zip(vecs, res):
        wxy, winf2 = ob.msm_g1(gm, None, v)
        ok = ok and binf == winf2 and np.array_equal(bxy, wxy)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,world", [(5000, 2), (3, 2), (2000, 4)])
def test_sharded_msm_gloo(n, world):
    """single and batched (one exchange for m MSMs) sharded MSM over world-size 2 and 4"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(r, True) for r in range(world)]
