#!/usr/bin/env python3
"""Re-entrancy stress of the C ABI (the reference calls MSM.compute from std.Thread workers, src/msm/mod.zig:637,732; a Zig host may also
hold several sessions at once): T host threads run random entry points for a while — MSMs on ONE shared wide-window handle (host scalars,
resident scalars, machine words, sub-ranges, batches), handles of their own that come and go (with and without the table), HyperKZG.open on
shared parameters, sumcheck sessions, runSumcheck, eq tables, product sessions — and every result is compared with a value the oracle
computed before the threads started. ctypes releases the GIL inside a call, so the calls really overlap. Run it under ZG_POOL_DEBUG=1 as
well: pooled blocks change hands between threads all the time.     usage: stress_threads.py [seconds=60] [threads=8] [seed=1]"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
from oracle import binding as ob  # noqa: E402  (the checker)
from tests import util as U  # noqa: E402
from zolt_amd import api, lib  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
lib.init(0)
N = 40000
GM = ob.g1_gen_multiples(N)
shared = lib.Bases.upload(GM)
params = api.HyperKZG.SetupParams(GM[:1 << 15], np.zeros(1 << 15, dtype=np.uint8))


def rand_fr(s, n):
    return ob.f_to_mont(ob.FR, U.random_raw256(s, n))


# expectations, per variant (computed once, single-threaded)
V = 4
sc = [rand_fr(9000 + i, N) for i in range(V)]
want_full = [ob.msm_g1(GM, None, s) for s in sc]
sub = [(1000 * (i + 1), 20000 + 777 * i) for i in range(V)]
want_sub = [ob.msm_g1(GM[o:o + m], None, sc[i][:m]) for i, (o, m) in enumerate(sub)]
words = [U.splitmix64(77 + i, N) >> np.uint64(i * 8) for i in range(V)]
want_words = [ob.msm_g1(GM, None, ob.f_from_u64(ob.FR, w)) for w in words]
own_n = [1500, 9000, 33000, 5003]
want_own = [ob.msm_g1(GM[:own_n[i]], None, sc[i][:own_n[i]]) for i in range(V)]
bat_n = 3000
want_bat = [ob.msm_g1(GM[:bat_n], None, sc[i][:bat_n]) for i in range(V)]
ev = [rand_fr(9100 + i, 1 << 14) for i in range(V)]
pt = [rand_fr(9200 + i, 14) for i in range(V)]
zero = np.zeros(4, dtype=np.uint64)
want_open = [ob.hyperkzg_open(GM[:1 << 15], np.zeros(1 << 15, dtype=np.uint8), ev[i], pt[i], zero) for i in range(V)]
want_run = [ob.run_sumcheck(ev[i][:1 << 12]) for i in range(V)]
want_eq = [ob.fr_eq_table(pt[i]) for i in range(V)]
want_halves = [ob.fr_sum_halves(ev[i]) for i in range(V)]
want_bind = [ob.fr_bind_low(ev[i][:1 << 10], pt[i][0]) for i in range(V)]
d_sc = [torch.from_numpy(s.view(np.int64)).cuda() for s in sc]
torch.cuda.synchronize()

errors, counts = [], [0] * T
stop = time.time() + budget


def same(a, b):
    return a[1] == b[1] and np.array_equal(a[0], b[0])


def worker(t):
    rng = np.random.default_rng(seed * 1000 + t)
    try:
        while time.time() < stop and not errors:
            i, op = int(rng.integers(0, V)), int(rng.integers(0, 11))
            if op == 0:
                ok = same(shared.msm(sc[i]), want_full[i])
            elif op == 1:
                o, m = sub[i]
                ok = same(shared.msm(sc[i][:m], off=o, n=m), want_sub[i])
            elif op == 2:
                ok = same(shared.msm_u64(words[i]), want_words[i])
            elif op == 3:
                ok = same(shared.msm_dev(d_sc[i].data_ptr(), N), want_full[i])
            elif op == 4:
                own = lib.Bases.upload(GM[:own_n[i]], None, expected_uses=int(rng.choice([0, 1])))
                ok = same(own.msm(sc[i][:own_n[i]]), want_own[i])
                own.free()
            elif op == 5:
                outs, infs = shared.msm_batch([sc[j][:bat_n] for j in range(V)], n=bat_n)
                ok = all(infs[j] == want_bat[j][1] and np.array_equal(outs[j], want_bat[j][0]) for j in range(V))
            elif op == 6:
                q, fin = api.HyperKZG.open(params, ev[i], pt[i], zero)
                wq, wqi, wfin = want_open[i]
                ok = np.array_equal(fin, wfin) and all(b == wqi[k] and np.array_equal(a, wq[k]) for k, (a, b) in enumerate(q))
            elif op == 7:
                r = lib.run_sumcheck(ev[i][:1 << 12])
                wc, wr, wch, wfin, wok = want_run[i]
                ok = r["result"] and np.array_equal(r["claim"], wc) and np.array_equal(r["final_eval"], wfin) and np.array_equal(r["rounds"].reshape(-1, 2, 4), np.asarray(wr).reshape(-1, 2, 4))
            elif op == 8:
                ok = np.array_equal(api.EqPolynomial(pt[i]).evals(), want_eq[i])
            elif op == 9:
                s = lib.SumcheckSession.open(ev[i])
                g0, g1 = s.round_sums()
                s.close()
                ok = np.array_equal(g0, want_halves[i][0]) and np.array_equal(g1, want_halves[i][1])
            else:
                p = api.DensePolynomial(ev[i][:1 << 10].copy())
                p.bindLow(pt[i][0])
                ok = np.array_equal(p.evaluations, want_bind[i])
            if not ok:
                errors.append(("mismatch", t, op, i))
            counts[t] += 1
    except Exception as e:  # noqa: BLE001
        errors.append((repr(e), t))


ths = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
t0 = time.time()
[th.start() for th in ths]
[th.join() for th in ths]
shared.free()
params.deinit()
st = lib.pool_debug_stats()
if errors:
    print("STRESS FAILED:", errors[:5])
    sys.exit(1)
print(f"stress ok: {sum(counts)} calls from {T} threads in {time.time() - t0:.1f} s, every result equal to the oracle's; pool debug: {st}")
