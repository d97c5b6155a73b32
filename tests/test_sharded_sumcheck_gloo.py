"""world_size-2/4 gloo tests of the sharded sumcheck orchestration (zolt_amd.api.ShardedSumcheck) on CPU.

The shard layout, the per-round 64-byte all-gather, the host-side modular sum, the residual gather and the
redundant tail rounds are the product's; the per-rank table operations are supplied by a test backend built on the
CPU oracle so the N > 1 control path runs without GPUs (on a GPU box the same class runs with
GpuSumcheckShardBackend — tests/test_gpu_sharded.py, bench.py --gpus N)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _OracleSession:
    def __init__(self, ob, table, layout):
        self.ob, self.t, self.layout = ob, np.ascontiguousarray(table, dtype=np.uint64).reshape(-1, 4), layout

    def round_sums(self):
        return self.ob.fr_sum_halves(self.t) if self.layout == 0 else self.ob.fr_sum_even_odd(self.t)

    def bind(self, r):
        self.t = self.ob.fr_bind_high(self.t, r) if self.layout == 0 else self.ob.fr_bind_low(self.t, r)

    def __len__(self):
        return self.t.shape[0]

    def final(self):
        return self.t[0].copy()

    def close(self):
        pass


class OracleSumcheckShardBackend:
    def __init__(self, local, layout):
        from oracle import binding as ob
        self.ob, self.layout = ob, layout
        self.s = _OracleSession(ob, local, layout)

    def local_len(self):
        return len(self.s)

    def round_sums(self):
        g0, g1 = self.s.round_sums()
        return torch.from_numpy(np.concatenate([g0, g1]).view(np.int64).copy())

    def bind(self, r):
        self.s.bind(r)

    def residual(self):
        return torch.from_numpy(self.s.t[0].view(np.int64).copy())

    def open_tail(self, table):
        return _OracleSession(self.ob, table, self.layout)

    def close(self):
        pass


def _worker(rank, world, port, v, layout, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import binding as ob
    from tests import util as U
    from zolt_amd import api
    n = 1 << v
    table = ob.f_to_mont(ob.FR, U.random_raw256(777 + v, n))
    # whole-table run with the oracle: sums, toy-verifier challenges, folds
    full = _OracleSession(ob, table, layout)
    g0, g1 = full.round_sums() if v else (table[0], np.zeros(4, dtype=np.uint64))
    claim = ob.f_add(ob.FR, g0, g1)
    ver_full, ver_sh = api.Sumcheck.Verifier(claim), api.Sumcheck.Verifier(claim)
    sh = api.ShardedSumcheck(OracleSumcheckShardBackend(table[api.sumcheck_shard_slice(n, world, rank, layout)], layout), world, rank)
    ok = True
    for _ in range(v):
        a, b = full.round_sums()
        want = np.stack([a, ob.f_sub(ob.FR, b, a)])
        got = sh.nextRound()
        ok &= bool(np.array_equal(got, want))
        ch_w, ch_g = ver_full.verifyRound(want), ver_sh.verifyRound(got)
        ok &= bool(np.array_equal(ch_w, ch_g))
        full.bind(ch_w)
        sh.receiveChallenge(ch_g)
    ok &= sh.isComplete() and bool(np.array_equal(sh.getFinalEval(), full.final()))
    ok &= bool(np.array_equal(ver_sh.claim, sh.getFinalEval()))
    # the rank's shard of the eq table = one ordinary eq-table build with the shared-prefix scalar
    r = ob.f_to_mont(ob.FR, U.random_raw256(99, v)) if v else np.zeros((0, 4), dtype=np.uint64)
    scale = ob.f_to_mont(ob.FR, U.random_raw256(5, 1))[0]
    eq_full = ob.fr_eq_table(r, scale)
    r_loc, sc_loc = api.sharded_eq_args(r, world, rank, layout, scale)
    ok &= bool(np.array_equal(ob.fr_eq_table(r_loc, sc_loc), eq_full[api.sumcheck_shard_slice(n, world, rank, layout)]))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,v,layout", [(2, 6, 0), (2, 6, 1), (4, 5, 0), (4, 2, 1), (2, 1, 0)])
def test_sharded_sumcheck_gloo(world, v, layout):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, v, layout, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(r, True) for r in range(world)]
