#!/usr/bin/env python3
"""CPU / GPU crossover per HOST-POINTER entry point: the smallest input size from which the GPU call (upload + kernels + download,
what an unmodified Zig call site pays through the shim) beats the reference's CPU body, restated in C (oracle/, single thread:
the reference's bodies are single-threaded loops, src/poly/mod.zig:160-175,252-290; src/msm/mod.zig:375-438).

    python tools/crossover.py [--out profiles/r3_crossover.json] [--max-log 20]

For n = 2^4 .. 2^max-log it times, on THIS box (GPU and host cores side by side):
    srs_commit     zg_msm_g1 on a resident SRS handle            vs  oracle msm_g1 (pippengerMSM / naive below 8 points)
    one_shot_msm   zg_g1_bases_upload(expected_uses=1) + zg_msm_g1 + free   vs  the same CPU MSM
    eq_table       zg_fr_eq_table (v = log2 n)                   vs  oracle fr_eq_table (evalsSliceWithScaling)
    bind_low       zg_fr_bind_low                                vs  oracle fr_bind_low
    bind_high      zg_fr_bind_high                               vs  oracle fr_bind_high
    run_sumcheck   zg_run_sumcheck (whole protocol on the device) vs  oracle run_sumcheck
    hyperkzg_open  zg_hyperkzg_open on the resident SRS          vs  oracle hyperkzg_open
    lt_table       zg_fr_lt_table (v = log2 n)                   vs  oracle lt_table (LtPolynomial.evaluateAtIndex per index)
    weighted_colsum zg_fr_weighted_colsum (sqrt n rows, two weight vectors) vs oracle weighted_colsum (the x_hi / x_lo double loop)
and reports per entry point the smallest power of two from which the GPU is faster at every larger measured size (`gate`). The
gates are the constants of zig/gpu/backend.zig (tests/test_abi_and_host.py holds the two against each other). The oracle is the
checker of the test suite; here it is only the CPU side of a timing comparison (tools/, never the product path)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402  (CPU side of the comparison)
from zolt_amd import lib  # noqa: E402


def best(fn, reps, budget_s=2.0):
    """least wall time of `reps` calls (at least one; stops early when the budget is spent)"""
    b, t_start = None, time.perf_counter()
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        b = dt if b is None else min(b, dt)
        if time.perf_counter() - t_start > budget_s:
            break
    return b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "crossover.json"), help="copy the file to profiles/ afterwards")
    ap.add_argument("--max-log", type=int, default=20)
    ap.add_argument("--cpu-budget", type=float, default=1.5, help="seconds of CPU work per (entry point, size); larger sizes are not timed on the CPU")
    ap.add_argument("--only", default="", help="comma-separated entry points to measure (default: all); the round-3 additions lt_table,weighted_colsum "
                    "were measured into profiles/r3_crossover_stage3.json this way")
    args = ap.parse_args()
    lib.init(0)
    nmax = 1 << args.max_log
    rng = np.random.default_rng(11)
    gm = ob.g1_gen_multiples(min(nmax, 1 << 16))
    if nmax > gm.shape[0]:  # the remaining bases on the device (the oracle's double-and-add would take minutes)
        ks = np.zeros((nmax, 4), dtype=np.uint64)
        ks[:, 0] = np.arange(1, nmax + 1, dtype=np.uint64)
        g = gm[0]
        gm, _ = lib.g1_scalar_mul_batch(np.repeat(g[None, :], nmax, axis=0), np.zeros(nmax, dtype=np.uint8), lib.field_op(lib.FR, lib.OP_TO_MONT, ks))
    sc = ob.f_to_mont(ob.FR, rng.integers(0, 1 << 63, size=(nmax, 4), dtype=np.uint64))
    srs = lib.Bases.upload(gm)
    one = np.zeros(4, dtype=np.uint64)
    res = {"box": {"cpu_cores": os.cpu_count(), "gpu": lib.version()}, "cpu_side": "oracle/zolt_oracle.c, single thread", "points": {}}
    logs = list(range(4, args.max_log + 1, 2)) + ([args.max_log] if args.max_log % 2 else [])
    skip_cpu = {}

    def measure(name, n, gpu_fn, cpu_fn, gpu_reps=20):
        gpu_fn()
        g = best(gpu_fn, gpu_reps)
        c = None
        if not skip_cpu.get(name):
            t0 = time.perf_counter()
            cpu_fn()
            first = time.perf_counter() - t0
            c = first if first > args.cpu_budget / 3 else min(first, best(cpu_fn, 5, args.cpu_budget))
            if first > args.cpu_budget:
                skip_cpu[name] = True  # every larger size costs the CPU more: stop timing it
        res["points"].setdefault(name, []).append({"n": n, "gpu_us": g * 1e6, "cpu_us": None if c is None else c * 1e6})
        print(f"{name:14s} n=2^{n.bit_length() - 1:<2d} gpu {g * 1e6:10.1f} us   cpu {'-' if c is None else format(c * 1e6, '10.1f')} us", flush=True)

    only = set(x for x in args.only.split(",") if x)
    _measure = measure

    def measure(name, *a, **kw):  # noqa: F811
        if not only or name in only:
            _measure(name, *a, **kw)

    for lg in logs:
        n = 1 << lg
        s_n, b_n = sc[:n], gm[:n]
        # round 3: LtPolynomial over the cube (ValEvaluationProver.init) and the Q-table / vector-matrix double loop (a square matrix, two suffixes)
        measure("lt_table", n, lambda: lib.fr_lt_table(sc[:lg]), lambda: ob.lt_table(sc[:lg]))
        rows = 1 << (lg // 2)
        measure("weighted_colsum", n, lambda: lib.fr_weighted_colsum(s_n, rows, n // rows, sc[:2 * rows].reshape(2, rows, 4)),
                lambda: ob.weighted_colsum(s_n, rows, n // rows, sc[:2 * rows].reshape(2, rows, 4)))
        if only and not (only - {"lt_table", "weighted_colsum"}):
            continue
        measure("srs_commit", n, lambda: srs.msm(s_n, 0, n), lambda: ob.msm_g1(b_n, None, s_n))

        def one_shot():
            h = lib.Bases.upload(b_n, expected_uses=1)
            h.msm(s_n)
            h.free()
        measure("one_shot_msm", n, one_shot, lambda: ob.msm_g1(b_n, None, s_n), gpu_reps=6)
        r = sc[:lg]
        measure("eq_table", n, lambda: lib.fr_eq_table(r), lambda: ob.fr_eq_table(r))
        measure("bind_low", n, lambda: lib.fr_bind_low(s_n, sc[5]), lambda: ob.fr_bind_low(s_n, sc[5]))
        measure("bind_high", n, lambda: lib.fr_bind_high(s_n, sc[5]), lambda: ob.fr_bind_high(s_n, sc[5]))
        measure("run_sumcheck", n, lambda: lib.run_sumcheck(s_n), lambda: ob.run_sumcheck(s_n))
        measure("hyperkzg_open", n, lambda: lib.hyperkzg_open(srs, s_n, sc[100:100 + lg], one), lambda: ob.hyperkzg_open(gm, None, s_n, sc[100:100 + lg], one), gpu_reps=6)
    srs.free()
    gates = {}
    for name, pts in res["points"].items():
        gate = None
        for p in reversed(pts):  # the smallest size from which the GPU wins at every larger measured size
            if p["cpu_us"] is not None and p["gpu_us"] >= p["cpu_us"]:
                break
            gate = p["n"]
        gates[name] = gate
    res["gates"] = gates
    res["note"] = ("gate = smallest measured n from which the GPU call is faster at every larger measured size (sizes step by 4x; "
                   "cpu_us null = not timed, the CPU was already far slower). zig/gpu/backend.zig's *_min constants are these values.")
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(gates))


if __name__ == "__main__":
    main()
