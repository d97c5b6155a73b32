"""ROUND-6 EXPERIMENT, NOT ADOPTED (the kernel is tools/exp/archive/r6_resident_sumcheck_kernel.patch; apply it to zolt_amd/csrc/poly.hip to re-run).
zg_run_sumcheck_dev (prover + toy verifier on the device) with the table resident in LDS across the chip (sc_run_resident_kernel, default
from 2^16 entries) against the launch-per-round protocol (ZG_SC_RESIDENT=0): python3 tools/exp/run_sumcheck_resident_ab.py [v ...]
Each mode runs in its own process (the switch is read once); transcripts are compared, then 200 calls are host-timed."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r"""
import hashlib, json, sys, time
import numpy as np
sys.path.insert(0, %r)
import torch
from zolt_amd import lib
lib.init(0)
out = {}
for v in %r:
    n = 1 << v
    rng = np.random.default_rng(v)
    ev = lib.field_op(lib.FR, lib.OP_TO_MONT, rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64))
    d = torch.from_numpy(ev.view(np.int64)).cuda()
    torch.cuda.synchronize()
    r = lib.run_sumcheck_dev(d.data_ptr(), n)
    h = hashlib.sha256(b"".join(np.ascontiguousarray(r[k]).tobytes() for k in ("claim", "rounds", "final_point", "final_eval"))).hexdigest()
    for _ in range(20):
        lib.run_sumcheck_dev(d.data_ptr(), n)
    t0 = time.perf_counter()
    for _ in range(200):
        lib.run_sumcheck_dev(d.data_ptr(), n)
    us = (time.perf_counter() - t0) / 200 * 1e6
    out[str(v)] = {"us": round(us, 1), "rounds_per_s": round(v / us * 1e6), "ok": bool(r["result"]), "sha": h[:16]}
print(json.dumps(out))
"""


def main():
    vs = [int(a) for a in sys.argv[1:]] or [16, 17, 18, 19, 20]
    res = {}
    for mode in ("1", "0"):
        env = dict(os.environ, ZG_SC_RESIDENT=mode)
        p = subprocess.run([sys.executable, "-c", CHILD % (ROOT, vs)], capture_output=True, text=True, env=env, timeout=600)
        if p.returncode:
            print(p.stderr[-2000:])
            return 1
        res["resident" if mode == "1" else "launch_per_round"] = json.loads(p.stdout.strip().splitlines()[-1])
    for v in vs:
        a, b = res["resident"][str(v)], res["launch_per_round"][str(v)]
        print(f"v = {v}: resident {a['us']} us ({a['rounds_per_s']} rounds/s), launch per round {b['us']} us ({b['rounds_per_s']} rounds/s), "
              f"same transcript: {a['sha'] == b['sha'] and a['ok'] and b['ok']}")
    print(json.dumps(res))
    return 0


if __name__ == "__main__":
    sys.exit(main())
