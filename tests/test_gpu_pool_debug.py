"""ZG_POOL_DEBUG (csrc/runtime.hip): the device pool's contract "a freed block is idle" is checked, not assumed. MSM.compute is called from
concurrent host threads (src/msm/mod.zig:355-372,637,732); a pooled block freed with work in flight and handed to another call would be a
silent wrong answer. Each case runs in a fresh process (the mode is read from the environment once)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import json, sys
import numpy as np
sys.path.insert(0, %r)
import torch
from zolt_amd import api, lib
from oracle import binding as ob
lib.init(0)
out = {"selftest": lib.pool_debug_selftest(), "after_selftest": lib.pool_debug_stats()}
# ordinary work through pooled scratch, sessions and handles, several host threads at once (the reference's calling pattern)
import threading
n = 5000
gm = ob.g1_gen_multiples(n)
rng = np.random.default_rng(5)
vecs = [ob.f_to_mont(ob.FR, rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)) for _ in range(4)]
want = [ob.msm_g1(gm, None, v) for v in vecs]
ok = [False] * 4
def work(i):
    good = True
    for _ in range(3):
        g, gi = api.MSM.compute(gm, vecs[i])            # upload + MSM + free: pooled workspaces come and go
        good = good and gi == want[i][1] and np.array_equal(g, want[i][0])
        ev = vecs[i][:1024]
        res = api.runSumcheck(api.DensePolynomial(ev))  # a pooled session
        wc, wr, wch, wfin, wok = ob.run_sumcheck(ev)
        good = good and res["result"] and np.array_equal(res["final_eval"], wfin) and np.array_equal(np.array(res["rounds"]), wr)
        good = good and np.array_equal(api.EqPolynomial(ev[:12]).evals(), ob.fr_eq_table(ev[:12]))
    ok[i] = bool(good)
ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
[t.start() for t in ts]
[t.join() for t in ts]
out["threads_ok"] = ok
out["stats"] = lib.pool_debug_stats()
print(json.dumps(out))
"""


def _run(mode):
    env = dict(os.environ)
    env.pop("ZG_POOL_DEBUG", None)
    env.pop("ZG_DEV_ALLOC_CACHE_MB", None)  # (0 = the pool keeps nothing: there is no reuse to check)
    if mode:
        env["ZG_POOL_DEBUG"] = str(mode)
    res = subprocess.run([sys.executable, "-c", SCRIPT % ROOT], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, (res.stdout[-500:], res.stderr[-1500:])
    return json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1]), res.stderr


def test_debug_mode_catches_a_block_written_after_its_free_and_clean_work_has_no_hits():
    out, err = _run(1)
    assert out["selftest"] == 1 and "ZG_POOL_DEBUG: block" in err and "written after it was freed" in err
    assert out["after_selftest"]["mode"] == 1 and out["after_selftest"]["hits"] == 0  # the self-test takes its own hit back
    assert all(out["threads_ok"])
    st = out["stats"]
    assert st["hits"] == 0 and st["blocks_verified"] > 10 and st["bytes_poisoned"] > 0, st


def test_mode_off_checks_nothing():
    out, err = _run(0)
    assert out["selftest"] == 0 and "ZG_POOL_DEBUG" not in err
    assert out["stats"] == {"mode": 0, "hits": 0, "suspect_frees": 0, "blocks_verified": 0, "bytes_poisoned": 0} and all(out["threads_ok"])
