#!/usr/bin/env python3
"""Stage-4 RegistersReadWriteChecking (zg_rrw_*) at a proving-size trace: the five 128 x 2^log_t tables built on the device, the
LOG_K + log_t rounds with the reference's phase split, per-phase times and the bytes a round moves.

    python tools/bench_stage4.py [--logt 20] [--p1 10]
Prints one JSON line. The algorithmic bytes of a round: the sums read ra and wa (2 tables) plus val where either is non-zero, the
fold reads and writes all five (5 reads + 2.5 writes of the live size): about 9.5 x live bytes per round, live = 128 * T_live * 32 B."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zolt_amd import api, lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--logt", type=int, default=20)
ap.add_argument("--p1", type=int, default=0, help="phase-1 rounds (default log_t / 2, the reference's split)")
args = ap.parse_args()
lib.init(0)
log_t = args.logt
p1 = args.p1 or log_t // 2
n = 1 << log_t
rng = np.random.default_rng(3)
ops = np.array((0x13, 0x03, 0x67, 0x1B, 0x33, 0x3B, 0x23, 0x63, 0x37, 0x6F, 0x17), dtype=np.uint32)
instr = rng.choice(ops, size=n) | (rng.integers(0, 32, size=n).astype(np.uint32) << 7) | (rng.integers(0, 32, size=n).astype(np.uint32) << 15) \
    | (rng.integers(0, 32, size=n).astype(np.uint32) << 20)
val = rng.integers(0, 1 << 63, size=n, dtype=np.uint64)
noop = np.zeros(n, dtype=bool)
r = lib.field_op(lib.FR, lib.OP_TO_MONT, rng.integers(0, 1 << 62, size=(2 * log_t + 8, 4), dtype=np.uint64))
gamma, r_cycle, chals = r[0], r[1:1 + log_t], r[1 + log_t:]
t0 = time.perf_counter()
cols = api.Stage4GruenProver.traceColumns(instr, val, noop, n)
t_cols = time.perf_counter() - t0
t0 = time.perf_counter()
p = api.Stage4GruenProver((instr, val, noop), gamma, r_cycle, p1, 7)
t_open = time.perf_counter() - t0
claim = api.fr_from_int(0)  # the timings do not depend on the claim; the parity tests hold the values
phases = {"phase1_cycle_gruen": [0.0, 0], "phase2_address": [0.0, 0], "phase3_cycle": [0.0, 0]}
rounds = []
t_all = time.perf_counter()
for k in range(7 + log_t):
    name = "phase1_cycle_gruen" if k < p1 else "phase2_address" if k < p1 + 7 else "phase3_cycle"
    live = p.current_K * p.current_T * 32
    t0 = time.perf_counter()
    ev = p.computeRoundEvals(k, claim)
    t1 = time.perf_counter()
    claim = api.cubicAtPoint(ev, chals[k])
    p.bindChallenge(k, chals[k])
    p._s.final()  # a 224-byte read on the session stream: waits for the folds
    t2 = time.perf_counter()
    phases[name][0] += t2 - t0
    phases[name][1] += 1
    rounds.append({"round": k, "live_MiB": live / 2**20, "sums_ms": (t1 - t0) * 1e3, "fold_ms": (t2 - t1) * 1e3,
                   "GBps_9.5x_of_stored": 9.5 * (live // 4 if k < p1 else live) / (t2 - t0) / 1e9})
total = time.perf_counter() - t_all
p.deinit()
print(json.dumps({"workload": f"stage4 registers read/write checking, K=128 x T=2^{log_t}, phases {p1}/7/{log_t - p1}",
                  "trace_columns_host_s": t_cols, "open_s_incl_columns": t_open, "rounds_total_ms": total * 1e3,
                  "phases_ms": {k: {"ms": v[0] * 1e3, "rounds": v[1]} for k, v in phases.items()},
                  "first_rounds": rounds[:4], "table_GiB_reference": 5 * 128 * n * 32 / 2**30, "table_GiB_stored": 5 * 32 * n * 32 / 2**30}))
