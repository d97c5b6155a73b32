"""RamReadWriteCheckingProver (src/zkvm/ram/read_write_checking.zig:160-1323) on the device session (zg_rwc_*) — SURVEY 8(f)3: the
three-phase sumcheck whose entry list is walked on the host (integers only) and evaluated / bound on the GPU next to its dense tables
(eq_evals, inc, val_init). Held against (1) the reference's own captured run, end to end
(tests/golden/rwc_captured_run.json: every printed prefix, the final claim and the three opening claims in full), and (2) the oracle's
restatement on random traces at 2^8, 2^13 and 2^20 cycles — every round polynomial, every bound entry, every claim, bit for bit."""
import json
import os
import random

import numpy as np
import pytest

from tests import util as U
from tests.test_transcript_host import check_rwc_against_the_captured_run

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from oracle import binding as ob
    from zolt_amd import api, lib
    lib.init()
    return api, lib, ob


def test_ram_read_write_checking_of_the_captured_run_on_the_device(env, golden_dir):
    api, lib, ob = env
    rwc = json.load(open(os.path.join(golden_dir, "rwc_captured_run.json")))
    stage2 = json.load(open(os.path.join(golden_dir, "stage2_batched_rounds.json")))
    elf = open(os.path.join(golden_dir, "fibonacci.elf"), "rb").read()
    accesses, gamma, r_cycle, initial_ram, challenges = U.rwc_inputs_of_the_captured_run(rwc, stage2, elf, api.fr_from_int)
    p = api.RamReadWriteCheckingProver(accesses, gamma, r_cycle, rwc["log_k"], rwc["log_t"], rwc["phase1_num_rounds"], rwc["start_address"],
                                       api.fr_from_int(0), initial_ram)
    try:
        check_rwc_against_the_captured_run(p, rwc, stage2, challenges, api.fr_to_int, last_q=lambda a: a.last_q)
    finally:
        p.deinit()


def _trace(seed, log_k, log_t, n_acc, start):
    """a random but CONSISTENT memory trace: reads return the last value written (or the initial one)"""
    rnd = random.Random(seed)
    K, T = 1 << log_k, 1 << log_t
    hot = [rnd.randrange(K) for _ in range(max(4, n_acc // 6))]  # few addresses, many accesses each: pairs, merges, checkpoints
    initial = {start + 8 * a: rnd.randrange(1 << 63) for a in rnd.sample(hot, len(hot) // 2)}
    mem = dict(initial)
    acc = []
    for ts in sorted(rnd.sample(range(T), min(n_acc, T))):
        a = start + 8 * rnd.choice(hot)
        if rnd.random() < 0.5:
            v = rnd.randrange(1 << 64)
            acc.append((ts, a, True, v))
            mem[a] = v
        else:
            acc.append((ts, a, False, mem.get(a, 0)))
    acc.append((T + 3, start, True, 1))            # beyond the trace length: ignored (:271)
    acc.append((1, start - 8, False, 0))           # below the RAM region: ignored (:273-279)
    acc.append((2, start + 8 * K, False, 0))       # beyond K words: ignored
    return acc, initial


@pytest.mark.parametrize("log_k,log_t,p1,n_acc", [(4, 8, 4, 200), (3, 8, 0, 256), (6, 8, 8, 90), (10, 13, 6, 3000), (16, 20, 10, 2500), (1, 1, 0, 2),
                                                  (5, 4, 2, 0), (12, 15, 7, 12000)])
@pytest.mark.parametrize("device_inc", [False, True])  # inc handed over (zg_rwc_open) / formed on the device from the write entries (zg_rwc_open_writes)
def test_ram_read_write_checking_vs_oracle(env, log_k, log_t, p1, n_acc, device_inc):
    api, lib, ob = env
    start = 0x80000000
    acc, initial = _trace(1000 * log_t + log_k, log_k, log_t, n_acc, start)
    gamma = ob.f_to_mont(ob.FR, U.random_raw256(77, 1))[0]
    gamma[:2] = 0  # a 128-bit challenge in its stored form, like the transcript's
    r_cycle = ob.f_to_mont(ob.FR, U.random_raw256(78 + log_t, log_t))
    o = ob.RamReadWriteCheckingProver(acc, gamma, r_cycle, log_k, log_t, p1, start, np.zeros(4, dtype=np.uint64), initial)
    # the true claim of the sum: sum_entries eq(r_cycle, cycle) * ra * (val + gamma * (val + inc(cycle)))
    P, g = ob._R_P, ob.fr_to_int(gamma)
    rows = sorted({e[0] for e in o.entries})
    eqv = {r: ob.fr_to_int(o.eq_evals[r]) for r in rows}
    incv = {r: ob.fr_to_int(o.inc[r]) for r in rows}
    claim = sum(eqv[e[0]] * e[2] * (e[3] + g * (e[3] + incv[e[0]])) for e in o.entries) % P
    o.current_claim = claim
    d = api.RamReadWriteCheckingProver(acc, gamma, r_cycle, log_k, log_t, p1, start, ob.fr_from_int(claim), initial, device_inc=device_inc)
    try:
        assert d.entry_list() == [(e[0], e[1], e[2]) for e in o.entries]
        chal = []
        for rd in range(log_k + log_t):
            we, ge = o.computeRoundPolynomialCubic(), d.computeRoundPolynomialCubic()
            assert np.array_equal(ge, we), rd
            assert (ob.fr_to_int(we[0]) + ob.fr_to_int(we[1])) % P == o.current_claim, rd  # and it is a sumcheck
            ch = ob.f_to_mont(ob.FR, U.random_raw256(5000 + rd, 1))[0]
            if rd % 2:
                ch[:2] = 0
                ch[3] &= np.uint64((1 << 61) - 1)
            chal.append(ch)
            for x in (o, d):
                x.updateClaim(we, ch)
                x.bindChallenge(ch)
            assert d.current_claim == o.current_claim and d.entry_list() == [(e[0], e[1], e[2]) for e in o.entries], rd
            if rd % 5 == 0 or rd >= log_k + log_t - 2:  # every field of every entry
                assert d.entries_full() == [[e[0], e[1], e[2] % P, e[3] % P, e[4], e[5]] for e in o.entries], rd
        assert d.isComplete() and o.isComplete()
        wo, go = o.getOpeningClaims(np.stack(chal)), d.getOpeningClaims(np.stack(chal))
        assert all(np.array_equal(a, b) for a, b in zip(go, wo))
    finally:
        d.deinit()


def test_two_writes_in_one_cycle_are_refused_by_the_write_list_form(env):
    """the reference keeps the LATER write of a cycle in access order, which the (cycle, address)-sorted list cannot tell:
    zg_rwc_open_writes refuses such a list and the dense form (zg_rwc_open) serves it"""
    api, lib, ob = env
    start = 0x80000000
    acc = [(3, start + 8, True, 5), (3, start, True, 9), (4, start, False, 9)]
    g = ob.fr_from_int(7)
    r_cycle = ob.f_to_mont(ob.FR, U.random_raw256(3, 3))
    with pytest.raises(RuntimeError, match="two writes in one cycle"):
        api.RamReadWriteCheckingProver(acc, g, r_cycle, 2, 3, 1, start, ob.fr_from_int(0), device_inc=True)
    d = api.RamReadWriteCheckingProver(acc, g, r_cycle, 2, 3, 1, start, ob.fr_from_int(0))
    o = ob.RamReadWriteCheckingProver(acc, g, r_cycle, 2, 3, 1, start, np.zeros(4, dtype=np.uint64))
    try:
        assert np.array_equal(d.computeRoundPolynomialCubic(), o.computeRoundPolynomialCubic())
    finally:
        d.deinit()


def test_gather_entry_points(env):
    api, lib, ob = env
    tab = ob.f_to_mont(ob.FR, U.random_raw256(9, 1 << 10))
    s = lib.SumcheckSession.open(tab, lib.SC_LOW_PAIR)
    idx = np.array([0, 1023, 5, 5, 512], dtype=np.uint64)
    assert np.array_equal(s.gather(idx), tab[idx.astype(np.int64)])
    r = ob.f_to_mont(ob.FR, U.random_raw256(10, 1))[0]
    s.bind(r)
    folded = ob.fr_bind_low(tab, r)
    assert np.array_equal(s.gather(np.array([511, 0], dtype=np.uint64)), folded[[511, 0]])
    with pytest.raises(lib.ZgError):
        s.gather(np.array([512], dtype=np.uint64))  # beyond the current length
    assert s.gather(np.zeros(0, dtype=np.uint64)).shape == (0, 4)
    s.close()
    p = lib.ProductSumcheckSession.open([tab, tab[::-1].copy()])
    assert np.array_equal(p.gather(1, idx), tab[::-1][idx.astype(np.int64)])
    p.bind(r)
    assert np.array_equal(p.gather(0, np.array([3], dtype=np.uint64)), folded[[3]])
    with pytest.raises(lib.ZgError):
        p.gather(2, idx)
    p.close()
