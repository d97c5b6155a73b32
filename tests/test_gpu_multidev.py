"""Several GPUs in ONE process behind the C ABI (include/zolt_gpu.h "several GPUs in one process"; csrc/sharded.hip): the
reference's own process model — ParallelMSM.compute runs threads inside `zolt prove` (src/msm/mod.zig:588-653),
ParallelBatchMSM (:683-748) / HyperKZG.batchCommit (src/poly/commitment/mod.zig:558-570) likewise.

The test box has ONE GPU, so the shard logic (partition, per-shard launch sets on worker threads, exchange, gather order,
combine) runs with several LOGICAL shards on that device (ZG_SHARDS, exchange by peer copies), and the RCCL exchange itself
(ncclCommInitAll + ncclAllGather through the dlopen'ed library) runs with the one shard RCCL accepts per device
(ZG_SHARD_EXCHANGE=rccl). Results must be byte-identical to the oracle's MSM(F,G).compute for every shard count."""
import numpy as np
import pytest

from tests import util as U

pytestmark = pytest.mark.gpu

N = 5000


@pytest.fixture(scope="module")
def env():
    from oracle import binding as ob
    from zolt_amd import api, lib
    lib.init()
    lib.init_devices(1)  # widen nothing on a 1-GPU box; on a multi-GPU box the tests still pin ZG_SHARDS explicitly
    gm = ob.g1_gen_multiples(N)
    return api, lib, ob, gm


def _rand(ob, seed, n):
    return ob.f_to_mont(ob.FR, U.random_raw256(seed, n))


@pytest.mark.parametrize("shards", [1, 2, 3, 8])
def test_sharded_msm_equals_oracle(env, shards, monkeypatch):
    api, lib, ob, gm = env
    monkeypatch.setenv("ZG_SHARDS", str(shards))
    inf = np.zeros(N, dtype=np.uint8)
    inf[11::97] = 1
    sb = lib.ShardedBases.upload(gm, inf)
    try:
        assert len(sb.shards()) == shards and sb.exchange() == ("none" if shards == 1 else "p2p")
        # the partition is ParallelMSM's: contiguous chunks of ceil(n / S) (src/msm/mod.zig:609,619-621)
        per = -(-N // shards)
        assert [(s, l) for _, s, l in sb.shards()] == [(min(i * per, N), max(0, min(per, N - i * per))) for i in range(shards)]
        sc = _rand(ob, 900 + shards, N)
        want = ob.msm_g1(gm, inf, sc)
        got = sb.msm(sc)
        assert got[1] == want[1] and np.array_equal(got[0], want[0])
        # equals the reference's ParallelMSM partition + serial combine, restated in the oracle
        wp = ob.msm_g1_parallel(gm, inf, sc, shards)
        assert got[1] == wp[1] and np.array_equal(got[0], wp[0])
        # a prefix (HyperKZG.commit of a shorter polynomial): trailing shards contribute the identity
        for n in sorted({0, 1, min(per, N), min(per + 1, N), N - 1}):
            w = ob.msm_g1(gm[:n], inf[:n], sc[:n])
            g = sb.msm(sc[:n], n)
            assert g[1] == w[1] and np.array_equal(g[0], w[0]), n
        # all-zero scalars -> identity from every shard (src/msm/mod.zig:949-966)
        z = sb.msm(np.zeros((N, 4), dtype=np.uint64))
        assert z[1] == 1 and not z[0].any()
    finally:
        sb.free()


@pytest.mark.parametrize("shards", [1, 2, 5])
@pytest.mark.parametrize("k,n", [(1, N), (4, N), (7, 1000), (3, 0)])
def test_sharded_batch_commit_equals_oracle(env, shards, k, n, monkeypatch):
    """E2: HyperKZG.batchCommit / ParallelBatchMSM sharded — k partials per shard in ONE exchange, k combines."""
    api, lib, ob, gm = env
    monkeypatch.setenv("ZG_SHARDS", str(shards))
    sb = lib.ShardedBases.upload(gm)
    try:
        batches = [_rand(ob, 1000 + 10 * k + j, n) for j in range(k)]
        out, inf = sb.msm_batch(batches, n)
        for j in range(k):
            w, wi = ob.msm_g1(gm[:n], None, batches[j])
            assert inf[j] == wi and np.array_equal(out[j], w), (shards, k, n, j)
    finally:
        sb.free()


def test_rccl_exchange_at_one_device(env, monkeypatch):
    """The RCCL leg itself: communicator creation (ncclCommInitAll via dlopen) and the grouped ncclAllGather, with the one rank
    per device RCCL accepts. Batch of 3 -> 288-byte gather."""
    api, lib, ob, gm = env
    monkeypatch.setenv("ZG_SHARDS", "1")
    monkeypatch.setenv("ZG_SHARD_EXCHANGE", "rccl")
    sb = lib.ShardedBases.upload(gm)
    try:
        assert sb.exchange() == "rccl"
        sc = _rand(ob, 1200, N)
        want = ob.msm_g1(gm, None, sc)
        got = sb.msm(sc)
        assert got[1] == want[1] and np.array_equal(got[0], want[0])
        batches = [_rand(ob, 1210 + j, N) for j in range(3)]
        out, inf = sb.msm_batch(batches)
        for j in range(3):
            w, wi = ob.msm_g1(gm, None, batches[j])
            assert inf[j] == wi and np.array_equal(out[j], w)
    finally:
        sb.free()


def test_communicator_sets_survive_rebinding_and_shutdown(env, monkeypatch):
    """Round-2 review: a 1-rank communicator created earlier in the process must not make a later, differently bound handle fail,
    and a handle that outlives zg_shutdown must keep a working communicator (csrc/sharded.hip: CommSet). On the one-GPU box the
    bound set cannot widen, so the sequence is: handle A (1 rank) -> second handle shares A's set -> zg_shutdown + zg_init_devices(1)
    -> handle B gets a NEW set while A still computes through its old one."""
    api, lib, ob, gm = env
    monkeypatch.setenv("ZG_SHARDS", "1")
    monkeypatch.setenv("ZG_SHARD_EXCHANGE", "rccl")
    sc = _rand(ob, 1250, N)
    want = ob.msm_g1(gm, None, sc)
    a = lib.ShardedBases.upload(gm)
    made = lib.sharded_comm_sets_created()
    assert made >= 1 and a.exchange() == "rccl"
    a2 = lib.ShardedBases.upload(gm[:64])
    assert lib.sharded_comm_sets_created() == made  # same number of bound devices: the set is shared
    a2.free()
    lib.shutdown()  # drops the library's reference to the set; A keeps its own
    lib.init()
    lib.init_devices(1)
    got = a.msm(sc)  # the review's out-of-bounds read: a live handle after zg_shutdown
    assert got[1] == want[1] and np.array_equal(got[0], want[0])
    b = lib.ShardedBases.upload(gm)
    assert lib.sharded_comm_sets_created() == made + 1  # re-created for the new binding, not an error
    for h in (a, b):
        got = h.msm(sc)
        assert got[1] == want[1] and np.array_equal(got[0], want[0])
    a.free()
    got = b.msm(sc)
    assert got[1] == want[1] and np.array_equal(got[0], want[0])
    b.free()


@pytest.mark.parametrize("shards,exchange", [(1, None), (1, "rccl"), (3, None)])
def test_pipelined_sharded_calls(env, shards, exchange, monkeypatch):
    """zg_msm_g1_sharded_dev_async / zg_msm_g1_batch_sharded_async + zg_sharded_wait: as many calls in flight as the handle has
    slots, waited for out of order; one more is refused; every result equals the oracle's."""
    api, lib, ob, gm = env
    monkeypatch.setenv("ZG_SHARDS", str(shards))
    if exchange:
        monkeypatch.setenv("ZG_SHARD_EXCHANGE", exchange)
    sb = lib.ShardedBases.upload(gm)
    try:
        R = sb.inflight()
        assert R == 3
        vecs = [_rand(ob, 1270 + j, N) for j in range(R)]
        wants = [ob.msm_g1(gm, None, v) for v in vecs]
        d = [[lib.DeviceBuffer.from_host(v[s:s + l]) for _, s, l in sb.shards()] for v in vecs]
        for rep in range(3):
            tickets = [sb.msm_dev_async([t.ptr for t in d[j]], N) for j in range(R)]
            with pytest.raises(lib.ZgError) as e:
                sb.msm_dev_async([t.ptr for t in d[0]], N)
            assert e.value.code == lib.ERR_INVALID and "zg_sharded_wait" in str(e.value)
            for j in ([1, 0, 2] if rep == 0 else reversed(range(R))):
                out, inf = sb.wait(tickets[j])
                assert inf[0] == wants[j][1] and np.array_equal(out[0], wants[j][0]), (rep, j)
            with pytest.raises(lib.ZgError):
                sb.wait(tickets[0])  # a ticket completes once
        # batches in flight beside a single MSM; a prefix; the synchronous form between them
        t0, k0, keep0 = sb.msm_batch_async(vecs)
        t1, k1, keep1 = sb.msm_batch_async([v[:777] for v in vecs[:2]], 777)
        t2 = sb.msm_dev_async([t.ptr for t in d[1]], N)
        out, inf = sb.wait(t1, k1)
        for j in range(2):
            w = ob.msm_g1(gm[:777], None, vecs[j][:777])
            assert inf[j] == w[1] and np.array_equal(out[j], w[0])
        got = sb.msm(vecs[2])  # takes the slot t1 freed
        assert got[1] == wants[2][1] and np.array_equal(got[0], wants[2][0])
        out, inf = sb.wait(t0, k0)
        for j in range(R):
            assert inf[j] == wants[j][1] and np.array_equal(out[j], wants[j][0])
        out, inf = sb.wait(t2)
        assert inf[0] == wants[1][1] and np.array_equal(out[0], wants[1][0])
    finally:
        sb.free()


def test_sharded_dev_call_ordered_behind_the_callers_stream(env, monkeypatch):
    """ready_streams: the shard's launch set waits for the caller's stream (advisor finding: the handle's streams are not ordered
    with whatever fills the device scalars). A long fill kernel chain on a torch stream, then the call without any synchronisation."""
    import torch
    api, lib, ob, gm = env
    monkeypatch.setenv("ZG_SHARDS", "2")
    sb = lib.ShardedBases.upload(gm)
    try:
        sc = _rand(ob, 1290, N)
        want = ob.msm_g1(gm, None, sc)
        dev = torch.device("cuda", 0)
        src = torch.from_numpy(sc.view(np.int64).copy()).to(dev)
        st = torch.cuda.Stream(device=dev)
        torch.cuda.synchronize()
        for rep in range(3):
            parts = [torch.zeros((l, 4), dtype=torch.int64, device=dev) for _, s, l in sb.shards()]
            torch.cuda.synchronize()
            with torch.cuda.stream(st):
                big = torch.empty(1 << 26, dtype=torch.int64, device=dev)
                for _ in range(8):
                    big.fill_(rep)  # ~0.5 GB per fill: the scalars below are written well after the call has returned
                for p_, (_, s, l) in zip(parts, sb.shards()):
                    p_.copy_(src[s:s + l])
            t = sb.msm_dev_async([p_.data_ptr() for p_ in parts], N, ready_streams=[st.cuda_stream] * len(parts))
            out, inf = sb.wait(t)
            assert inf[0] == want[1] and np.array_equal(out[0], want[0]), rep
    finally:
        sb.free()


def test_sharded_msm_resident_scalars(env, monkeypatch):
    api, lib, ob, gm = env
    monkeypatch.setenv("ZG_SHARDS", "4")
    sb = lib.ShardedBases.upload(gm)
    try:
        sc = _rand(ob, 1300, N)
        d_parts = [lib.DeviceBuffer.from_host(sc[s:s + l]) for _, s, l in sb.shards()]
        got = sb.msm_dev([t.ptr for t in d_parts], N)
        want = ob.msm_g1(gm, None, sc)
        assert got[1] == want[1] and np.array_equal(got[0], want[0])
    finally:
        sb.free()


def test_api_parallel_msm_mirrors(env, monkeypatch):
    api, lib, ob, gm = env
    monkeypatch.setenv("ZG_SHARDS", "3")
    sc = _rand(ob, 1400, 3000)
    got = api.ParallelMSM.compute(gm[:3000], sc)
    want = ob.msm_g1(gm[:3000], None, sc)
    assert got[1] == want[1] and np.array_equal(got[0], want[0])
    batches = [_rand(ob, 1410 + j, 3000) for j in range(2)]
    out, inf = api.ParallelBatchMSM.compute(gm[:3000], batches)
    for j in range(2):
        w, wi = ob.msm_g1(gm[:3000], None, batches[j])
        assert inf[j] == wi and np.array_equal(out[j], w)


def test_hyperkzg_commit_and_batch_commit_on_a_sharded_srs(env, monkeypatch):
    """north_star's "HyperKZG batch-commit shard ... across the GPUs" at the mirror level: SetupParams(sharded=True) keeps one
    shard of the SRS per device; commit / batchCommit (equal lengths fused into one sharded batch call, others one by one)
    equal the oracle's HyperKZG.commit (src/poly/commitment/mod.zig:239-255,558-570)."""
    api, lib, ob, gm = env
    monkeypatch.setenv("ZG_SHARDS", "4")
    n = 2048
    inf = np.zeros(n, dtype=np.uint8)
    params = api.HyperKZG.SetupParams(gm[:n], inf, sharded=True)
    try:
        polys = [_rand(ob, 1800 + k, n) for k in range(3)] + [_rand(ob, 1810, 700), np.zeros((0, 4), dtype=np.uint64), _rand(ob, 1811, 3000)]
        got = api.HyperKZG.batchCommit(params, polys)
        for (xy, fl), p in zip(got, polys):
            w, wi = ob.hyperkzg_commit(gm[:n], inf, p)
            assert fl == wi and np.array_equal(xy, w)
        c, ci = api.HyperKZG.commit(params, polys[0])
        w, wi = ob.hyperkzg_commit(gm[:n], inf, polys[0])
        assert ci == wi and np.array_equal(c, w)
    finally:
        params.deinit()


@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("shards,v", [(1, 6), (2, 6), (4, 9), (8, 3), (8, 12), (3, 7)])
def test_sharded_sumcheck_session_equals_single_device(env, layout, shards, v, monkeypatch):
    """zg_sumcheck_*_sharded: every round message, and the final evaluation, equal the oracle's for the whole table — LOW_PAIR
    shards by contiguous chunks, HIGH_HALF by residue class; 8 shards of a 2^3 table start directly in the tail; 3 devices
    use 2 shards (largest power of two)."""
    api, lib, ob, gm = env
    monkeypatch.setenv("ZG_SHARDS", str(shards))
    table = _rand(ob, 1500 + 16 * shards + v, 1 << v)
    s = lib.ShardedSumcheckSession.open(table, layout)
    try:
        want_shards = 1
        while want_shards * 2 <= min(shards, 1 << v):
            want_shards *= 2
        assert s.shards() == want_shards
        cur = table
        for rd in range(v):
            g0, g1 = s.round_sums()
            w0, w1 = (ob.fr_sum_halves(cur) if layout == lib.SC_HIGH_HALF else ob.fr_sum_even_odd(cur))
            assert np.array_equal(g0, w0) and np.array_equal(g1, w1), rd
            r = _rand(ob, 1600 + rd, 1)[0]
            s.bind(r)
            cur = ob.fr_bind_high(cur, r) if layout == lib.SC_HIGH_HALF else ob.fr_bind_low(cur, r)
            assert len(s) == len(cur)
        assert np.array_equal(s.final(), cur[0])
    finally:
        s.close()


def _n_gpus():
    from zolt_amd import lib
    return lib.device_count()


def _several_shards_body(lib, ob, gm, n_shards, exchange, devices):
    """the body of the >= 2-GPU test, parametrised by where the shards live: single MSM, prefixes, a batch, the pipelined entry points,
    and the sharded sumcheck session in both layouts"""
    sb = lib.ShardedBases.upload(gm)
    try:
        assert len(sb.shards()) == n_shards and sb.exchange() == exchange
        assert sorted(d for d, _, _ in sb.shards()) == devices
        sc = _rand(ob, 1700, N)
        for n in (N, N // 2 + 1, 3):
            want = ob.msm_g1(gm[:n], None, sc[:n])
            got = sb.msm(sc[:n], n)
            assert got[1] == want[1] and np.array_equal(got[0], want[0]), n
        batches = [_rand(ob, 1710 + j, N) for j in range(5)]
        out, inf = sb.msm_batch(batches)
        for j in range(5):
            w, wi = ob.msm_g1(gm, None, batches[j])
            assert inf[j] == wi and np.array_equal(out[j], w)
        # several calls in flight on the handle's slots (several slots' exchanges on one communicator)
        tickets = [sb.msm_batch_async([batches[j]]) for j in range(min(sb.inflight(), 5))]
        for j, (t, k, keep) in enumerate(tickets):
            o, fl = sb.wait(t, k)
            w, wi = ob.msm_g1(gm, None, batches[j])
            assert fl[0] == wi and np.array_equal(o[0], w), j
    finally:
        sb.free()
    for layout in (lib.SC_HIGH_HALF, lib.SC_LOW_PAIR):
        table = _rand(ob, 1720 + layout, 1 << 12)
        s = lib.ShardedSumcheckSession.open(table, layout)
        try:
            assert s.shards() > 1
            cur = table
            for rd in range(12):
                g0, g1 = s.round_sums()
                w0, w1 = (ob.fr_sum_halves(cur) if layout == lib.SC_HIGH_HALF else ob.fr_sum_even_odd(cur))
                assert np.array_equal(g0, w0) and np.array_equal(g1, w1), rd
                r = _rand(ob, 1730 + rd, 1)[0]
                s.bind(r)
                cur = ob.fr_bind_high(cur, r) if layout == lib.SC_HIGH_HALF else ob.fr_bind_low(cur, r)
            assert np.array_equal(s.final(), cur[0])
        finally:
            s.close()


@pytest.mark.skipif(_n_gpus() < 2, reason="needs at least two GPUs in this box (the real RCCL all-gather over xGMI)")
def test_real_devices_rccl_all_gather(env, monkeypatch):
    """On a multi-GPU box: one shard per physical device, partials exchanged by the grouped ncclAllGather, combined on device 0 —
    single MSM, prefix, batch, pipelined calls, and the sharded sumcheck session across the devices. Skipped on the one-GPU test box;
    test_two_logical_shards_same_body below runs the same body there."""
    api, lib, ob, gm = env
    monkeypatch.delenv("ZG_SHARDS", raising=False)
    monkeypatch.delenv("ZG_SHARD_EXCHANGE", raising=False)
    nd = min(_n_gpus(), 8)
    lib.init_devices(nd)
    assert lib.n_devices() >= nd
    _several_shards_body(lib, ob, gm, lib.n_devices(), "rccl", list(range(lib.n_devices())))


def test_two_logical_shards_same_body(env, monkeypatch):
    """The body of the >= 2-GPU test with ZG_SHARDS = 2 on ONE device: every branch of csrc/sharded.hip that does not need a second
    physical GPU runs — partition, per-shard worker threads and launch sets, slot pipeline, gather order, combine, the sharded session's
    local rounds and its residual rounds — with the partials exchanged by peer copies. What remains unexecuted on a one-GPU box is
    exactly the collective among SEVERAL communicator ranks: `ncclGroupStart .. ncclAllGather x S .. ncclGroupEnd` with S > 1 in
    the EX_RCCL branch of csrc/sharded.hip's exchange step (its S = 1 form runs in test_rccl_exchange_at_one_device) and ncclCommInitAll over
    more than one device (comms_acquire)."""
    api, lib, ob, gm = env
    monkeypatch.setenv("ZG_SHARDS", "2")
    monkeypatch.delenv("ZG_SHARD_EXCHANGE", raising=False)
    _several_shards_body(lib, ob, gm, 2, "p2p", [0, 0])


def test_sharded_entry_points_reject_bad_arguments(env, monkeypatch):
    """Error behaviour of the several-GPU entry points: same codes and messages as their single-device counterparts
    (ZG_ERR_INVALID, never a crash, never a silent wrong answer)."""
    api, lib, ob, gm = env
    monkeypatch.setenv("ZG_SHARDS", "3")
    sb = lib.ShardedBases.upload(gm[:100])
    try:
        with pytest.raises(lib.ZgError) as e:
            sb.msm(_rand(ob, 1, 101), 101)  # range exceeds the uploaded bases
        assert e.value.code == lib.ERR_INVALID and "exceeds" in str(e.value)
        with pytest.raises(lib.ZgError):
            sb.msm_batch([_rand(ob, 2, 101)], 101)
        out, inf = sb.msm_batch([], 0)
        assert out.shape == (0, 8)
        z = sb.msm(np.zeros((0, 4), dtype=np.uint64), 0)  # n = 0 -> identity (src/msm/mod.zig:361-363)
        assert z[1] == 1 and not z[0].any()
    finally:
        sb.free()
    with pytest.raises(lib.ZgError) as e:
        lib.ShardedSumcheckSession.open(_rand(ob, 3, 12), lib.SC_LOW_PAIR)  # length not a power of two
    assert e.value.code == lib.ERR_INVALID
    s = lib.ShardedSumcheckSession.open(_rand(ob, 4, 1), lib.SC_LOW_PAIR)  # a single element: already complete
    assert len(s) == 1 and s.shards() == 1
    with pytest.raises(lib.ZgError):
        s.bind(_rand(ob, 5, 1)[0])
    assert np.array_equal(s.final(), _rand(ob, 4, 1)[0])
    s.close()
    monkeypatch.setenv("ZG_SHARD_EXCHANGE", "rccl")  # RCCL needs exactly one shard per bound device
    with pytest.raises(lib.ZgError) as e:
        lib.ShardedBases.upload(gm[:100])
    assert e.value.code == lib.ERR_INVALID and "one shard per bound device" in str(e.value)
