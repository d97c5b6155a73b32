"""Sharded MSM and sharded sumcheck with the real GPU backends (zolt_amd.api.Gpu*ShardBackend).

The GPU box has one device, so the N > 1 cases run several ranks on cuda:0 with the gloo rendezvous (records staged
through the host); the world-size-1 case runs the RCCL (`nccl`) collective path in-process. Both are compared with the
oracle bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_case(rank, world, v, layout, n_msm):
    """One rank's share of: sharded eq table -> sharded sumcheck, and a sharded MSM. Returns True on parity."""
    from oracle import binding as ob
    from tests import util as U
    from zolt_amd import api, lib as zl
    dev = torch.device("cuda", 0)
    ok = True
    work = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(work):
        n = 1 << v
        # --- the rank's shard of eq(r, .) built on the device with the shared-prefix scalar, then a sharded sumcheck of it
        r = ob.f_to_mont(ob.FR, U.random_raw256(31 + v, v))
        scale = ob.f_to_mont(ob.FR, U.random_raw256(8, 1))[0]
        eq_full = ob.fr_eq_table(r, scale)
        sl = api.sumcheck_shard_slice(n, world, rank, layout)
        r_loc, sc_loc = api.sharded_eq_args(r, world, rank, layout, scale)
        d_tab = torch.empty((n // world, 4), dtype=torch.int64, device=dev)
        zl.fr_eq_table_dev(r_loc, d_tab.data_ptr(), scale=sc_loc, stream=work.cuda_stream)
        ok &= bool(np.array_equal(d_tab.cpu().numpy().view(np.uint64), eq_full[sl]))
        sh = api.ShardedSumcheck(api.GpuSumcheckShardBackend(d_tab, layout), world, rank)
        table = eq_full
        g0, g1 = ob.fr_sum_halves(table) if layout == 0 else ob.fr_sum_even_odd(table)
        ver_w, ver_g = api.Sumcheck.Verifier(ob.f_add(ob.FR, g0, g1)), api.Sumcheck.Verifier(ob.f_add(ob.FR, g0, g1))
        for _ in range(v):
            a, b = ob.fr_sum_halves(table) if layout == 0 else ob.fr_sum_even_odd(table)
            want = np.stack([a, ob.f_sub(ob.FR, b, a)])
            got = sh.nextRound()
            ok &= bool(np.array_equal(got, want))
            ch = ver_w.verifyRound(want)
            ok &= bool(np.array_equal(ch, ver_g.verifyRound(got)))
            table = ob.fr_bind_high(table, ch) if layout == 0 else ob.fr_bind_low(table, ch)
            sh.receiveChallenge(ch)
        ok &= sh.isComplete() and bool(np.array_equal(sh.getFinalEval(), table[0]))
        sh.deinit()
        # --- sharded MSM (ParallelMSM partition, un-normalised partials, device combine)
        gm = ob.g1_gen_multiples(n_msm)
        sc = ob.f_to_mont(ob.FR, U.random_raw256(4242, n_msm))
        s, e = api.shard_bounds(n_msm, world)[rank]
        bases = zl.Bases.upload(gm[s:e]) if e > s else zl.Bases.upload(gm[:1])
        d_sc = torch.from_numpy(np.ascontiguousarray(sc[s:e]).view(np.int64).copy()).to(dev)
        msm = api.ShardedMSM(api.GpuShardBackend(bases, e - s), world, rank)
        xy, inf = msm.compute(d_sc)
        want, winf = ob.msm_g1(gm, None, sc)
        ok &= inf == winf and bool(np.array_equal(xy, want))
        bases.free()
    return bool(ok)


def _worker(rank, world, port, v, layout, n_msm, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok = _run_case(rank, world, v, layout, n_msm)
    except Exception as ex:  # report instead of hanging the peers
        ok = repr(ex)
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,v,layout,n_msm", [(2, 12, 0, 5000), (4, 11, 1, 4097)])
def test_sharded_ranks_share_one_gpu(world, v, layout, n_msm):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, v, layout, n_msm, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
    assert res == [(r, True) for r in range(world)]


@pytest.mark.parametrize("layout", [0, 1])
def test_sharded_world1_rccl(layout):
    """world size 1 over the RCCL backend: the device-tensor all-gather path (`all_gather_into_tensor`) end to end."""
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
    try:
        assert _run_case(0, 1, 10, layout, 3000)
    finally:
        if created:
            dist.destroy_process_group()
