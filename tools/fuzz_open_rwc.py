#!/usr/bin/env python3
"""Randomised differential test of two composite paths against the CPU oracle (not part of the pytest suite):
  * HyperKZG.open — random table sizes 2^8 .. 2^18 on SRS handles of random length (shorter than the table: the first commits are
    clamped; infinity bases), points with fewer variables than the table, every ZG_HK_FUSE_LONG mode (the fused long launch set whose
    sort walks the rows by their live lengths, the split form, one launch set per level), a resident or a host table;
  * RamReadWriteCheckingProver — random consistent memory traces, random phase split, inc handed over or formed on the device: every
    round polynomial, every bound entry, the opening claims (the cycle-phase walk runs on the device, the address phase on the host).
usage: fuzz_open_rwc.py [seconds=120] [seed=1]"""
import os
import random
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import binding as ob  # noqa: E402  (the checker)
from tests import util as U  # noqa: E402
from zolt_amd import api, lib  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rnd = random.Random(seed)
lib.init(0)
GM = ob.g1_gen_multiples(1 << 18)
P = ob._R_P


def rand_fr(s, n):
    return ob.f_to_mont(ob.FR, U.random_raw256(s, n))


# Ambient code-path switches (docs/design/08_switches.md), drawn per iteration BEFORE the handle is made and kept for its lifetime (some are
# read when the handle is planned, some at every launch). Round 6: a whole-suite run under ZG_MSM_TWO_PASS_SORT=0 found a fused launch set
# that could not sort; the fuzzers now walk these switches too. Most iterations keep the defaults.
AMBIENT = {"ZG_MSM_TWO_PASS_SORT": ["0"], "ZG_MSM_LDS_SORT": ["0"], "ZG_MSM_REDUCE_2D": ["0"], "ZG_MSM_ALONE_FULL": ["0"], "ZG_MSM_BATCH_FUSE": ["0"],
           "ZG_MSM_SIDE_TABLE": ["0"], "ZG_MSM_FINE_BITS": ["5", "6"], "ZG_MSM_FINE_BITS_MIN": ["7", "8"], "ZG_MSM_HOST_AFFINE": ["0"],
           "ZG_MSM_ROWCOL_WAVE_FROM": ["0", "1"], "ZG_MSM_TABLE_SPAN_MB": ["8", "32"], "ZG_MSM_LANES": ["1", "2"], "ZG_MSM_ROWS_SHARED_TAIL": ["0"]}


def draw_ambient():
    for k in AMBIENT:
        os.environ.pop(k, None)
    picked = {}
    if rnd.random() < 0.4:
        for k in rnd.sample(sorted(AMBIENT), rnd.choice([1, 1, 2, 3])):
            picked[k] = rnd.choice(AMBIENT[k])
            os.environ[k] = picked[k]
    return picked


def one_open(it):
    ambient = draw_ambient()
    v = rnd.choice([8, 10, 12, 14, 15, 16, 16, 17, 17, 18])
    srs_n = rnd.choice([1 << v, 1 << (v - 1), (1 << (v - 1)) + rnd.randrange(1, 1000), 1 << max(v - 2, 4), min(1 << 18, 1 << (v + 1))])
    srs_n = min(srs_n, 1 << 18)
    inf = np.zeros(srs_n, dtype=np.uint8)
    if rnd.random() < 0.5:
        inf[rnd.randrange(0, 50)::rnd.randrange(100, 3000)] = 1
    os.environ["ZG_HK_FUSE_LONG"] = str(rnd.choice([1, 1, 2, 0]))
    params = api.HyperKZG.SetupParams(GM[:srs_n], inf)
    try:
        ev = rand_fr(100000 + it, 1 << v)
        if rnd.random() < 0.3:  # a sparse witness column: zeros and small values
            mask = np.array([rnd.random() < 0.7 for _ in range(64)] * ((1 << v) // 64), dtype=bool)
            ev[mask] = 0
        nv = v if rnd.random() < 0.6 else rnd.randrange(1, v + 1)
        pt = rand_fr(200000 + it, v)[:nv]
        zero = np.zeros(4, dtype=np.uint64)
        wq, wqi, wfin = ob.hyperkzg_open(GM[:srs_n], inf, ev, pt, zero)
        if rnd.random() < 0.5:
            quotients, final = api.HyperKZG.open(params, ev, pt, zero)
            q = np.stack([a for a, _ in quotients]) if quotients else np.zeros((0, 8), dtype=np.uint64)
            qi = np.array([b for _, b in quotients], dtype=np.uint8)
        else:
            d_ev = lib.DeviceBuffer.from_host(ev)
            q, qi, final = lib.hyperkzg_open_dev(params._dev, d_ev.ptr, 1 << v, pt, zero)
        assert np.array_equal(final, wfin), ("open final", v, srs_n, nv, os.environ["ZG_HK_FUSE_LONG"], ambient)
        assert np.array_equal(np.asarray(qi, dtype=np.uint8), wqi) and np.array_equal(q, wq), ("open quotients", v, srs_n, nv, os.environ["ZG_HK_FUSE_LONG"], ambient)
    finally:
        params.deinit()


def trace(log_k, log_t, n_acc, start):
    K, T = 1 << log_k, 1 << log_t
    hot = [rnd.randrange(K) for _ in range(max(2, rnd.choice([n_acc // 6, n_acc // 40 + 2, 3])))]
    initial = {start + 8 * a: rnd.randrange(1 << 63) for a in rnd.sample(hot, len(hot) // 2)}
    mem = dict(initial)
    acc = []
    for ts in sorted(rnd.sample(range(T), min(n_acc, T))):
        a = start + 8 * rnd.choice(hot)
        if rnd.random() < 0.5:
            val = rnd.randrange(1 << 64)
            acc.append((ts, a, True, val))
            mem[a] = val
        else:
            acc.append((ts, a, False, mem.get(a, 0)))
    return acc, initial


def one_rwc(it):
    log_t = rnd.choice([1, 3, 6, 8, 8, 10, 12, 13])
    log_k = rnd.choice([1, 3, 4, 6, 10, 12])
    p1 = rnd.randrange(0, log_t + 1)
    n_acc = rnd.choice([0, 2, 50, 300, 1500, 6000])
    start = 0x80000000
    acc, initial = trace(log_k, log_t, n_acc, start)
    gamma = rand_fr(300000 + it, 1)[0]
    gamma[:2] = 0
    r_cycle = rand_fr(400000 + it, log_t)
    device_inc = rnd.random() < 0.5
    o = ob.RamReadWriteCheckingProver(acc, gamma, r_cycle, log_k, log_t, p1, start, np.zeros(4, dtype=np.uint64), initial)
    g = ob.fr_to_int(gamma)
    rows = sorted({e[0] for e in o.entries})
    eqv = {r: ob.fr_to_int(o.eq_evals[r]) for r in rows}
    incv = {r: ob.fr_to_int(o.inc[r]) for r in rows}
    claim = sum(eqv[e[0]] * e[2] * (e[3] + g * (e[3] + incv[e[0]])) for e in o.entries) % P
    o.current_claim = claim
    try:
        d = api.RamReadWriteCheckingProver(acc, gamma, r_cycle, log_k, log_t, p1, start, ob.fr_from_int(claim), initial, device_inc=device_inc)
    except RuntimeError as e:  # two writes in one cycle are refused by the write-list form: documented
        assert device_inc and "two writes in one cycle" in str(e), e
        return
    tag = ("rwc", log_k, log_t, p1, n_acc, device_inc)
    try:
        chal = []
        for rd in range(log_k + log_t):
            we, ge = o.computeRoundPolynomialCubic(), d.computeRoundPolynomialCubic()
            assert np.array_equal(ge, we), tag + (rd,)
            ch = rand_fr(500000 + 100 * it + rd, 1)[0]
            if rd % 2:
                ch[:2] = 0
            chal.append(ch)
            for x in (o, d):
                x.updateClaim(we, ch)
                x.bindChallenge(ch)
            if rd % 3 == 0 or rd >= log_k + log_t - 2:
                assert d.entries_full() == [[e[0], e[1], e[2] % P, e[3] % P, e[4], e[5]] for e in o.entries], tag + (rd, "entries")
        wo, go = o.getOpeningClaims(np.stack(chal)), d.getOpeningClaims(np.stack(chal))
        assert all(np.array_equal(a, b) for a, b in zip(go, wo)), tag + ("claims",)
    finally:
        d.deinit()


t0 = time.time()
n_open = n_rwc = 0
it = 0
while time.time() - t0 < budget:
    it += 1
    if rnd.random() < 0.4:
        one_open(it)
        n_open += 1
    else:
        one_rwc(it)
        n_rwc += 1
print("fuzz ok: %d HyperKZG.open calls, %d RamReadWriteChecking provers in %.1f s" % (n_open, n_rwc, time.time() - t0))
