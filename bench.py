#!/usr/bin/env python3
"""bench.py — BN254 G1 MSM/sec on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--logn 20]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    (without a launcher, `python bench.py --gpus N` starts its N ranks itself, before anything touches a GPU)

A step = one pass of the hot path over one batch of synthetic input: --msms-per-step (default 32) independent 2^logn-point MSMs
(value = MSMs per second = steps x msms_per_step / time), each over device-resident bases P_i = (i+1)·G (the reference bench's
point family, src/bench.zig:261-268) and device-resident uniform scalars (splitmix64, seed
0x5A4F4C54; several scalar vectors rotate across steps). N > 1: the point/scalar arrays are
sharded in ParallelMSM's contiguous chunks (src/msm/mod.zig:609), each rank computes its Jacobian
partial, the partials are all-gathered over RCCL and combined on the device (strong scaling).

Rank 0 prints ONE compact JSON line (< 4 KB, the LAST line of stdout: compact_line below) with the driver contract fields plus
  roofline     — dominant kernel (msm_accumulate): algorithmic bytes / HIP-event kernel time vs 8 TB/s
  cpu_baseline — the C restatement of the reference's pippengerMSM timed on this box's host cores
  also         — scalars only: the 2^22 figure, the table-less figure, sumcheck rounds/s (config 3) with their roofline fractions
Everything else (per-kernel times, notes, the prove path, the sharded legs: the old `extra` object) goes to the side file
bench_extra.json in the cwd (copy: profiles/bench_extra_last.json), like the reference's harness, which prints one short result per size
(src/bench.zig:243-287). `--full-line` (internal: the child processes of the extras) prints the full object instead.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# HBM traffic and executed VALU instructions of one msm_accumulate launch come from a COMMITTED counter file that
# tools/collect_profiles.sh writes on the GPU box (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU in passes of their own, serial
# bench): nothing is typed in by hand here, and tests/test_abi_and_host.py checks that what this file reports is what the JSON holds.
# Calibration on this access pattern (tools/microbench gather|stream under --pmc FETCH_SIZE, the file's "calibration" object): random
# 64-byte rows read with the kernel's load shape are reported at 1.04x their bytes (taken as counted), a sequential 16 B/lane stream at
# 0.50x (the gfx950 x2 of MI355X_MICROARCH.md). A launch gathers one 64-byte table row per addition and streams 4 bytes of sorted
# reference per addition (counted at half: + 0.5 * 4 * additions).
PMC_FILE = os.path.join(ROOT, "profiles", "r6_pmc.json")


def load_pmc(path=PMC_FILE):
    """{'2^20': {'FETCH_SIZE': KB, 'WRITE_SIZE': KB, 'SQ_INSTS_VALU': n, ...}, '2^22': {...}} per launch of msm_accumulate, or {}"""
    try:
        with open(path) as fh:
            d = json.load(fh)
        return {size: {k: v["avg"] for k, v in blk.get("counters", {}).items()} for size, blk in d.get("sizes", {}).items()}
    except (OSError, ValueError, KeyError):
        return {}


def profiled_duration_ms(logn, path=PMC_FILE):
    """the kernel's average duration in the committed rocprofv3 --kernel-trace --stats run of the serial bench (ms), or None"""
    try:
        with open(path) as fh:
            d = json.load(fh)
        return d["sizes"][f"2^{logn}"]["duration"]["avg_us"] / 1e3
    except (OSError, ValueError, KeyError, TypeError):
        return None


def measured_traffic(pmc, logn, adds_per_launch):
    """HBM bytes of one msm_accumulate launch from the committed counters (None when the file has no entry for this size)"""
    c = pmc.get(f"2^{logn}", {})
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        return None
    return c["FETCH_SIZE"] * 1024.0 + 0.5 * 4.0 * adds_per_launch + c["WRITE_SIZE"] * 1024.0


TRAFFIC_SOURCE = ("rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE (separate passes, serial bench) per msm_accumulate launch, read from profiles/r6_pmc.json "
                  "(written by tools/collect_profiles.sh; the text summaries of the same run are profiles/r6_rocprofv3_summary*.txt): FETCH_SIZE taken as "
                  "counted for the gathered 64-byte rows, + 0.5 x 4 B x additions for the streamed sorted references (gfx950 tallies a wide stream at "
                  "half its bytes; calibrated with tools/microbench gather|stream, same file), + WRITE_SIZE")
# static instruction mix of one lazy-limb XYZZ mixed add (hipcc --save-temps of the accumulate fast path): 1467
# v_mad_u64_u32 + 146 v_lshl_add_u64 + 144 v_lshrrev_b64 + 81 v_mul_lo_u32 at 4 issue cycles per wave, 382 32-bit
# add/and/shift/sub at 2 (tools/microbench.hip rates)
MADD_ISSUE_CYCLES = (1467 + 146 + 144 + 81) * 4 + 382 * 2
VALU_PEAK_GCYC = 1024 * 2.4  # 256 CUs x 4 SIMDs x 2.4 GHz
# the same mix priced with the issue times MEASURED on MI355X at 2 waves per SIMD (profiles/r1_microbench_instruction_rates.txt,
# ns per wave-instruction per SIMD under load: v_mad_u64_u32 2.034, v_lshl_add_u64 2.181 (v_lshrrev_b64 taken equal), v_mul_lo_u32
# 2.113, 32-bit add 1.183): the least time one SIMD needs for one wave-wide mixed add, clocks as they really are under this load
MADD_MIN_NS_PER_SIMD = 1467 * 2.034 + (146 + 144) * 2.181 + 81 * 2.113 + 382 * 1.183
# The executed instruction count comes from the counters instead of the static mix (round-3 review): SQ_INSTS_VALU per msm_accumulate
# launch, rocprofv3 --pmc in a pass of its own, read from PMC_FILE — about 2382 wave instructions per wave-wide mixed add against the
# 2220 of the static fast path; the 162 on top are the loop around the add (sorted-reference decode, row address, conditional negation of
# y, chunk bookkeeping), 32-bit work priced at the simple-op rate.
MADD_STATIC_INSTRS = 1467 + 146 + 144 + 81 + 382
SIMPLE_OP_NS = 1.183
SEED = 0x5A4F4C54
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
N_SCALAR_SETS = 3


def splitmix64(seed, n):
    idx = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def raw_scalars(seed, start, count):
    """rows [start, start+count) of the (n,4) raw 256-bit stream of splitmix64(seed)"""
    idx = np.arange(4 * start + 1, 4 * (start + count) + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z.reshape(count, 4)


def closed_form_scalar(raw, start):
    """(sum_i raw_i * (start+i+1)) mod r with exact integers (SURVEY §8(d) self-check)."""
    from zolt_amd.api import R_MOD
    k = np.arange(start + 1, start + raw.shape[0] + 1, dtype=object)
    tot = 0
    for limb in range(4):
        tot += int((raw[:, limb].astype(object) * k).sum()) << (64 * limb)
    return tot % R_MOD


COMPACT_LIMIT = 4096  # bytes: the driver's parser reads the last stdout line; round 5's 22 KB line outgrew its capture window


def _r(x, digits=6):
    """floats at 6 significant digits (the line is a summary; the side file keeps full precision)"""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    return x


def _get(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def _json_default(o):
    """numpy scalars / arrays that found their way into an object: their Python value (never a TypeError at the very end of a run)"""
    if hasattr(o, "tolist"):
        return o.tolist()
    return str(o)


def compact_line(out, extra_file="bench_extra.json"):
    """The line the driver parses: the contract's keys, `config` / `roofline` / `cpu_baseline` reduced to numbers and short names, and the
    north_star's other figures as scalars under `also`. Built from the full object `out` (which goes to `extra_file` unshortened)."""
    cfg, roof, cpu, ex = out.get("config", {}), out.get("roofline", {}), out.get("cpu_baseline"), out.get("extra", {})
    be = cfg.get("breakeven_msms") or {}
    line = {k: _r(out.get(k)) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                        "vs_baseline", "dtype", "data")}
    line["config"] = {k: _r(cfg.get(k)) for k in ("workload", "points", "points_per_gpu", "msms_per_step", "ms_per_msm", "streams", "window_bits",
                                                   "windows", "table_levels", "table_build_ms", "table_bytes")}
    line["config"]["mode"] = cfg.get("mode")
    line["config"]["breakeven_msms"] = [_r(be.get("pipelined")), _r(be.get("one_at_a_time"))]  # [pipelined, one at a time]
    line["config"]["collective_ranks"] = cfg.get("collective_ranks")
    line["config"]["sharding"] = "single GPU" if out.get("n_gpus", 1) == 1 else "contiguous chunks, all-gather of 96 B partials, device combine"
    line["config"]["bit_exact_check"] = "closed form via scalarMul, every timed MSM"
    line["roofline"] = {k: _r(roof.get(k)) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch",
                                                      "launches_per_msm", "avg_launch_ms", "rocprofv3_avg_launch_ms", "frac_at_rocprofv3_duration")}
    line["roofline"]["valu_issue_frac"] = _r(_get(roof, "valu_issue_measured_rates", "frac_from_pmc_count") or _get(roof, "valu_issue", "frac"))
    line["roofline"]["counters"] = os.path.relpath(PMC_FILE, ROOT) if roof.get("traffic") is not None else None
    line["roofline"]["binds"] = "VALU issue (integer Fp products), not HBM"
    if cpu is not None:
        line["cpu_baseline"] = {k: _r(cpu.get(k)) for k in ("value", "unit", "cores", "kind", "sample", "seconds_per_msm", "result_checked",
                                                             "host_cores_available")}
        line["cpu_baseline"]["seconds_per_msm_2e22"] = _r(_get(cpu, "sizes", "2^22", "seconds_per_msm"))
        line["cpu_baseline"]["parallel_msm_per_s"] = _r(_get(cpu, "parallel_msm", "value"))
        line["cpu_baseline"]["parallel_threads"] = _get(cpu, "parallel_msm", "threads")
        line["cpu_baseline"]["sumcheck_v20_rounds_per_s"] = _r(_get(cpu, "sumcheck_v20", "rounds_per_s"))
    sc = ex.get("sumcheck_v20", {})
    also = {
        "msm_2e22_per_s": _r(_get(ex, "msm_2^22_single_gpu", "value")), "msm_2e22_ms": _r(_get(ex, "msm_2^22_single_gpu", "ms_per_msm")),
        "msm_2e22_roofline_frac": _r(_get(ex, "msm_2^22_single_gpu", "roofline", "frac")),
        "msm_table_less_per_s": _r(_get(ex, "msm_no_precompute", "value")), "msm_table_less_ms": _r(_get(ex, "msm_no_precompute", "ms_per_msm")),
        "msm_host_scalars_ms": _r(ex.get("msm_host_scalars_ms")),
        "sumcheck_rounds_per_s": _r(sc.get("rounds_per_s")), "sumcheck_ms": _r(sc.get("ms_per_sumcheck")),
        "sumcheck_roofline_frac": _r(_get(sc, "roofline", "frac")),
        "sumcheck_device_resident_rounds_per_s": _r(_get(ex, "sumcheck_v20_device_resident", "rounds_per_s")),
        "sumcheck_device_resident_roofline_frac": _r(_get(ex, "sumcheck_v20_device_resident", "roofline", "frac")),
        "sumcheck_host": "compiled C++ loop over the C ABI" if "python_binding" in sc else "python binding",
        "eq_table_roofline_frac": _r(_get(sc, "kernel_rooflines", "eq_main_kernel", "frac")),
        "fold_round0_roofline_frac": _r(_get(sc, "kernel_rooflines", "sc_fold_kernel_round0", "frac")),
        "fold_2e24_TBps": _r(_get(ex, "fold_2^24_by_challenge_kind", "narrow_TBps")),
        "hyperkzg_open_v20_resident_ms": _r(_get(ex, "hyperkzg_open", "v20_resident_ms")),
        "prove_path_ms": _r(_get(ex, "prove_path", "total_ms")), "prove_path_single_use_key_ms": _r(_get(ex, "prove_path_single_use_key", "total_ms")),
        "key_2e24_setup_ms": _r(_get(ex, "largest_key_2^24+256", "table", "setup_ms")), "key_2e24_table_bytes": _get(ex, "largest_key_2^24+256", "table", "table_bytes"),
        "key_2e24_commit_ms": _r(_get(ex, "largest_key_2^24+256", "table", "commit_ms")),
        "key_2e24_table_less_commit_ms": _r(_get(ex, "largest_key_2^24+256", "table_less", "commit_ms")),
        "msm_2e22_sharded_per_s": _r(_get(ex, "msm_2^22_sharded", "value")),
        "sumcheck_sharded_rounds_per_s": _r(_get(ex, "sumcheck_v20_sharded", "rounds_per_s")),
    }
    line["also"] = {k: v for k, v in also.items() if v is not None}
    line["extra_file"] = extra_file
    errs = {k: str(v)[:120] for k, v in (("extras", ex.get("extras_error")), ("cpu_baseline", out.get("cpu_baseline_error"))) if v}
    if errs:  # a leg behind the headline failed: the line says so instead of silently lacking a key
        line["errors"] = errs
    if out.get("provisional"):
        line["provisional"] = True  # N > 1 only: printed right after the timed region; the last line of a complete run supersedes it
    text = json.dumps(line, separators=(",", ":"), default=_json_default)
    if len(text) >= COMPACT_LIMIT:  # cannot happen with the fixed key set above; never let a long string take the line down
        line.pop("also")
        text = json.dumps(line, separators=(",", ":"), default=_json_default)
    if len(text) >= COMPACT_LIMIT:  # still too long (a pathological string somewhere): the contract's keys and the two objects' numbers only
        line["config"] = {k: v for k, v in line["config"].items() if not isinstance(v, str) or len(v) <= 40}
        line["roofline"] = {k: v for k, v in line["roofline"].items() if not isinstance(v, str) or len(v) <= 40}
        if "cpu_baseline" in line:
            line["cpu_baseline"] = {k: v for k, v in line["cpu_baseline"].items() if not isinstance(v, str) or len(v) <= 40}
        text = json.dumps(line, separators=(",", ":"), default=_json_default)
    return text


def write_side_file(out, name="bench_extra.json"):
    """the full object, unshortened: cwd/bench_extra.json and profiles/bench_extra_last.json (rank 0 only). Returns the paths written."""
    paths = []
    for path in (os.path.join(os.getcwd(), name), os.path.join(ROOT, "profiles", "bench_extra_last.json")):
        try:
            with open(path, "w") as fh:
                json.dump(out, fh, indent=1, default=_json_default)
            paths.append(path)
        except (OSError, TypeError, ValueError):  # the side file is a convenience: it must never stand between a finished run and its line
            pass
    return paths


def emit(out, full_line=False):
    """rank 0's last act: side file, then the one line (flushed; nothing is printed after it)"""
    if full_line:
        print(json.dumps(out, default=_json_default), flush=True)
        return
    write_side_file(out)
    sys.stdout.flush()
    try:  # RCCL prints its version banner through C stdio; on a pipe that buffer would reach stdout at exit, AFTER the line: flush it first
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    print(compact_line(out), flush=True)


RANK_LOG_DIR = os.environ.get("ZOLT_BENCH_LOG_DIR") or os.path.join(ROOT, "gpurun_out", "bench_ranks")
STALL_LIMIT_S = float(os.environ.get("ZOLT_BENCH_STALL_S", "120"))


class Watchdog:
    """A rank that makes no progress for STALL_LIMIT_S seconds must not hang the job (an RCCL collective that never completes waits for
    ever): a daemon thread in every rank watches a heartbeat the main thread advances at every phase and step; on a stall it writes where
    the rank stood (phase, all thread tracebacks) to RANK_LOG_DIR/rank<r>.log and to stderr and ends THIS process with exit code 3 —
    torch.distributed.run (or self_launch below) then takes the other ranks down. Nothing is re-executed; the heartbeat file
    rank<r>.beat is also what the self-launcher polls."""

    def __init__(self, rank, world):
        import threading
        self.rank, self.world, self.phase, self.limit = rank, world, "start", STALL_LIMIT_S
        self.child = None  # the extras' running child process (run_child), ended before this process is
        self.last = time.monotonic()
        self.log = self.beat_path = None
        try:
            os.makedirs(RANK_LOG_DIR, exist_ok=True)
            self.log = open(os.path.join(RANK_LOG_DIR, f"rank{rank}.log"), "w")
            self.beat_path = os.path.join(RANK_LOG_DIR, f"rank{rank}.beat")
        except OSError:
            pass
        self.note(f"rank {rank} of {world}, pid {os.getpid()}, argv {sys.argv[1:]}")
        threading.Thread(target=self._watch, daemon=True).start()

    def note(self, msg):
        if self.log:
            self.log.write(f"[{time.strftime('%H:%M:%S')}] {msg}\n")
            self.log.flush()

    def beat(self, phase=None, limit=None):
        """progress: the stall clock restarts; `limit` widens it for one known-long phase (a child process, the CPU baseline)"""
        if phase is not None and phase != self.phase:
            self.phase = phase
            self.note(phase)
        self.limit = limit if limit is not None else STALL_LIMIT_S
        self.last = time.monotonic()
        if self.beat_path:
            try:
                with open(self.beat_path, "w") as fh:
                    fh.write(f"{time.time():.3f} {self.limit:.0f} {self.phase}\n")
            except OSError:
                pass

    def _watch(self):
        import faulthandler
        while True:
            time.sleep(2.0)
            idle = time.monotonic() - self.last
            if idle > self.limit:
                msg = f"bench.py rank {self.rank}: no progress for {idle:.0f} s in phase '{self.phase}' (limit {self.limit:.0f} s): giving up"
                for fh in (self.log, sys.stderr):
                    if fh:
                        try:
                            fh.write(msg + "\n")
                            faulthandler.dump_traceback(file=fh, all_threads=True)
                            fh.flush()
                        except (OSError, ValueError):
                            pass
                ch = self.child
                if ch is not None and ch.poll() is None:
                    ch.kill()  # exactly the child this process started
                os._exit(3)


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one FRESH process per GPU, the env torch.distributed.run would
    set, rendezvous on 127.0.0.1), relay rank 0's output, keep every rank's stderr (and the other ranks' stdout) in RANK_LOG_DIR, and
    watch them: when a rank exits non-zero, or its heartbeat (Watchdog) goes stale, the launcher kills exactly the children it
    started and exits non-zero with a JSON line that names the rank. Nothing in this parent process ever touches a GPU."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.makedirs(RANK_LOG_DIR, exist_ok=True)
    for r in range(n):
        try:
            os.remove(os.path.join(RANK_LOG_DIR, f"rank{r}.beat"))
        except OSError:
            pass
    procs, files = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), ZOLT_BENCH_LOG_DIR=RANK_LOG_DIR)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        err = open(os.path.join(RANK_LOG_DIR, f"rank{r}.stderr"), "w")
        out = None if r == 0 else open(os.path.join(RANK_LOG_DIR, f"rank{r}.stdout"), "w")
        files += [f for f in (err, out) if f]
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out, stderr=err))
    t_start, failed = time.time(), None
    while failed is None and any(pr.poll() is None for pr in procs):
        time.sleep(1.0)
        for r, pr in enumerate(procs):
            rc = pr.poll()
            if rc not in (None, 0):
                failed = (r, f"exit code {rc}")
                break
            if rc is None:
                try:
                    with open(os.path.join(RANK_LOG_DIR, f"rank{r}.beat")) as fh:
                        stamp, limit, phase = fh.read().split(None, 2)
                    if time.time() - float(stamp) > float(limit) + 15.0:  # the rank's own watchdog should have fired: it is wedged below Python
                        failed = (r, f"heartbeat stale for {time.time() - float(stamp):.0f} s in phase '{phase.strip()}'")
                        break
                except (OSError, ValueError):
                    if time.time() - t_start > STALL_LIMIT_S + 60.0:  # never got as far as its first heartbeat
                        failed = (r, "no heartbeat since launch")
                        break
    for pr in procs:
        if failed is not None and pr.poll() is None:
            pr.kill()  # exactly the processes started above
    rc = 0
    for pr in procs:
        rc = max(rc, abs(pr.wait()))
    for f in files:
        f.close()
    if failed is not None:
        r, why = failed
        tail = ""
        try:
            with open(os.path.join(RANK_LOG_DIR, f"rank{r}.stderr")) as fh:
                tail = fh.read()[-1500:]
        except OSError:
            pass
        print(json.dumps({"error": f"bench.py --gpus {n}: rank {r} failed ({why}); all ranks stopped", "rank": r, "logs": RANK_LOG_DIR, "stderr_tail": tail}))
        return rc or 3
    return rc


def single_process_mode(args):
    """The reference's own process model: ONE process drives --single-process GPUs through the C ABI (zg_init_devices,
    zg_g1_bases_upload_sharded, zg_msm_g1_sharded_dev: per-device Pippenger on worker threads, ONE ncclAllGather of 96 B per
    device, combine on device 0). Prints one JSON object; called by the main bench as a child process."""
    import torch
    from zolt_amd import api, lib
    nd = args.single_process
    lib.init(0)
    lib.init_devices(nd)
    n = 1 << args.logn
    g = api.generator()
    ks = np.zeros((n, 4), dtype=np.uint64)
    ks[:, 0] = np.arange(1, n + 1, dtype=np.uint64)
    bases_xy, _ = lib.g1_scalar_mul_batch(np.repeat(g[None, :], n, axis=0), np.zeros(n, dtype=np.uint8), lib.field_op(lib.FR, lib.OP_TO_MONT, ks))
    res = {"devices": nd, "points": n}
    for mode in (["auto"] if nd > 1 else ["auto", "rccl"]):
        if mode != "auto":
            os.environ["ZG_SHARD_EXCHANGE"] = mode
        sb = lib.ShardedBases.upload(bases_xy)
        shards = sb.shards()
        raw = raw_scalars(SEED, 0, n)
        sm = lib.field_op(lib.FR, lib.OP_TO_MONT, raw)
        parts = [torch.from_numpy(sm[s:s + l].view(np.int64).copy()).to(torch.device("cuda", d)) for d, s, l in shards]
        ptrs = [t.data_ptr() for t in parts]
        want = api.MSM.scalarMul(g, api.fr_from_int(closed_form_scalar(raw, 0)))
        for _ in range(3):
            got = sb.msm_dev(ptrs, n)
        assert got[1] == want[1] and np.array_equal(got[0], want[0]), "sharded MSM result mismatch"
        reps = 30
        t0 = time.perf_counter()
        for _ in range(reps):
            sb.msm_dev(ptrs, n)
        el = (time.perf_counter() - t0) / reps
        res[sb.exchange()] = {"ms_per_msm": el * 1e3, "msm_per_s": 1.0 / el, "shards": len(shards),
                              "note": "synchronous calls (host result per MSM), scalars resident per device"}
        # the pipelined entry points: as many calls in flight as the handle has slots, results collected in order
        inflight, tickets, reps = sb.inflight(), [], 60
        t0 = time.perf_counter()
        for _ in range(reps):
            if len(tickets) == inflight:
                o, fl = sb.wait(tickets.pop(0))
            tickets.append(sb.msm_dev_async(ptrs, n))
        while tickets:
            o, fl = sb.wait(tickets.pop(0))
        el = (time.perf_counter() - t0) / reps
        assert fl[0] == want[1] and np.array_equal(o[0], want[0]), "pipelined sharded MSM result mismatch"
        res[sb.exchange()]["pipelined"] = {"ms_per_msm": el * 1e3, "msm_per_s": 1.0 / el, "in_flight": inflight,
                                           "note": "zg_msm_g1_sharded_dev_async + zg_sharded_wait"}
        h = sb.msm(sm)  # host-scalar entry point: per-device H2D of the shard on the worker threads
        assert h[1] == want[1] and np.array_equal(h[0], want[0])
        t0 = time.perf_counter()
        for _ in range(5):
            sb.msm(sm)
        res[sb.exchange()]["host_scalars_ms_per_msm"] = (time.perf_counter() - t0) / 5 * 1e3
        sb.free()
    print(json.dumps(res))
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--logn", type=int, default=20)
    ap.add_argument("--msms-per-step", type=int, default=32,
                    help="independent MSMs (scalar vectors) per step; 20 steps x 32 MSMs keep the timed region near a second")
    ap.add_argument("--single-process", type=int, default=0,
                    help="internal: measure the one-process / several-GPU C ABI (zg_init_devices + zg_msm_g1_sharded_dev) on this many devices")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-2e22", action="store_true",
                    help="measure the single-thread CPU baseline at 2^22 points IN FULL (~45 s) and write profiles/cpu_baseline_2e22_full.json")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--full-line", action="store_true",
                    help="internal (child processes of the extras): print the full object on one line instead of the compact line + side file")
    ap.add_argument("--window-bits", type=int, default=0)
    ap.add_argument("--precompute", type=int, default=0)
    ap.add_argument("--streams", type=int, default=0,
                    help="independent MSMs are issued round-robin on this many HIP streams (1 = strictly serial; 0 = 3 on one GPU, 4 per rank "
                         "when the work is sharded: measured best at 2^17..2^19 points per rank, tools/ab_sharded_streams.sh)")
    args = ap.parse_args()

    if args.single_process:
        return single_process_mode(args)
    if args.cpu_baseline_2e22:
        return cpu_baseline_2e22_full()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus)  # no launcher: become the launcher (nothing has touched a GPU yet)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    fault = os.environ.get("ZOLT_BENCH_FAULT", "")  # supervision self-test (tests/test_bench_supervision.py, CPU): "stall:<rank>" / "exit:<rank>"
    if fault and int(fault.split(":")[1]) == rank:
        if fault.startswith("exit"):
            raise SystemExit(7)
        Watchdog(rank, world).beat("injected stall")
        time.sleep(3600)
    elif fault:
        Watchdog(rank, world).beat("healthy rank waiting on a peer that never arrives", limit=3600)
        time.sleep(3600)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} (or without a launcher)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libzolt_gpu has no CPU fallback)")
    wd = Watchdog(rank, world)  # a stalled rank ends itself after STALL_LIMIT_S: a hung collective must not hang the driver
    # one process per GPU; ZOLT_BENCH_DIST_BACKEND=gloo lets several ranks share one GPU to exercise the
    # sharded path on a 1-GPU box (RCCL refuses two ranks on one device) — never used for reported numbers
    dist_backend = os.environ.get("ZOLT_BENCH_DIST_BACKEND", "nccl")
    if dist_backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # ZOLT_BENCH_FORCE_SHARDED=1 (with torchrun --nproc-per-node 1) drives the sharded code path — partial MSM,
    # RCCL all-gather, device combine — on a single GPU, so it can be exercised on a 1-GPU box
    force_sharded = bool(os.environ.get("ZOLT_BENCH_FORCE_SHARDED")) and "WORLD_SIZE" in os.environ
    use_dist = world > 1 or force_sharded
    if use_dist:
        if dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(dist_backend)

    wd.beat("process group up")
    from zolt_amd import api, lib
    lib.init(local_rank)
    wd.beat("library initialised")

    # all work runs on an explicit (non-default) torch stream: the C ABI treats a NULL stream as "the library's own
    # stream", which is not ordered with torch's legacy default stream (handle 0)
    work_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(work_stream)
    stream = work_stream.cuda_stream
    assert stream != 0
    nstreams = args.streams if args.streams > 0 else (4 if use_dist else 3)
    if world > 1 and dist_backend != "nccl":
        nstreams = 1  # the gloo debugging path stages through the host and is synchronous anyway
    tstreams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=dev) for _ in range(nstreams - 1)]
    g = api.generator()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def run_size(logn, steps, warmup, per_step):
        """time `steps` steps of `per_step` MSMs of 2^logn points (sharded over the ranks), every result checked against the closed form"""
        n = 1 << logn
        bounds = api.shard_bounds(n, world)
        start, end = bounds[rank]
        n_loc = end - start

        # ---- synthetic inputs, generated with the product's own kernels (untimed)
        wd.beat(f"2^{logn}: generating inputs")
        t0 = time.time()
        ks = np.zeros((n_loc, 4), dtype=np.uint64)
        ks[:, 0] = np.arange(start + 1, end + 1, dtype=np.uint64)
        ks_m = lib.field_op(lib.FR, lib.OP_TO_MONT, ks)
        bases_xy, bases_inf = lib.g1_scalar_mul_batch(np.repeat(g[None, :], n_loc, axis=0), np.zeros(n_loc, dtype=np.uint8), ks_m)
        assert not bases_inf.any()
        d_bases = torch.from_numpy(bases_xy.view(np.int64)).to(dev)
        torch.cuda.synchronize()
        tb0 = time.perf_counter()  # the handle's one-time cost (untimed in the metric, priced in config.table_build_ms): plan, workspaces,
        bases = lib.Bases.upload_dev(d_bases.data_ptr(), 0, n_loc, stream=stream, window_bits=args.window_bits,  # the table of multiples
                                     precompute_levels=args.precompute)
        torch.cuda.synchronize()
        table_build_ms = (time.perf_counter() - tb0) * 1e3
        raws, d_scalars, expect_k = [], [], []
        for s in range(N_SCALAR_SETS):
            raw = raw_scalars(SEED + s, start, n_loc)
            sm = lib.field_op(lib.FR, lib.OP_TO_MONT, raw)  # reduces mod r and converts, like F.fromBytes
            raws.append(raw)
            d_scalars.append(torch.from_numpy(sm.view(np.int64)).to(dev))
            expect_k.append(closed_form_scalar(raw, start))
        setup_s = time.time() - t0
        wd.beat(f"2^{logn}: inputs resident, table built")

        backend = api.GpuShardBackend(bases, n_loc)
        sharded = api.ShardedMSM(backend, world, rank)
        d_res = torch.zeros((max(steps, warmup, 1) * per_step, 9), dtype=torch.int64, device=dev)  # xy[8] + flag word

        def step(i):
            """one step = per_step independent MSMs (scalar vectors rotate), issued round-robin on the work streams. Sharded: each
            stream's share of the step goes through ONE exchange (ShardedMSM.compute_batch: m partials, one all-gather of m * 96 bytes
            per rank, one combine launch) instead of one 96-byte collective per MSM."""
            if not use_dist:
                for b in range(per_step):
                    j = i * per_step + b
                    sc = d_scalars[j % N_SCALAR_SETS]
                    st = tstreams[j % nstreams].cuda_stream
                    bases.msm_dev_async(sc.data_ptr(), n_loc, d_res[j].data_ptr(), d_res[j, 8:].data_ptr(), stream=st)
                return
            ns = min(nstreams, per_step)
            for s in range(ns):
                lo, hi = i * per_step + s * per_step // ns, i * per_step + (s + 1) * per_step // ns
                if hi == lo:
                    continue
                with torch.cuda.stream(tstreams[s]):
                    sharded.compute_batch([d_scalars[j % N_SCALAR_SETS] for j in range(lo, hi)], out=d_res[lo:hi])

        for i in range(warmup):
            step(i)
            wd.beat(f"2^{logn}: warmup")
        barrier()
        wd.beat(f"2^{logn}: timed region")

        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        barrier()
        elapsed = time.perf_counter() - t0
        wd.beat(f"2^{logn}: timed region done")  # (no heartbeat inside the timed region: a file write per step would be in it)

        # outside the timed region (no instrumentation inside it): the same issue pattern once more with HIP-event brackets around
        # every kernel group — the durations of kernels that SHARE the GPU with the other streams' work ("overlapped", transparency only)
        lib.profile_begin(8 * 64 + 64)
        for i in range(min(steps, max(1, 64 // per_step))):
            step(i)
        barrier()
        prof = lib.profile_end()
        wd.beat(f"2^{logn}: profiled legs")

        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist_backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

        # outside the timed region: the same MSM strictly serial on one stream, so that each kernel's duration is its own
        # (in the timed region kernels of different streams share the GPU and their HIP-event durations stretch)
        d_tmp = torch.zeros(9, dtype=torch.int64, device=dev)
        lib.profile_begin(8 * 6 + 64)
        for i in range(6):
            bases.msm_dev_async(d_scalars[i % N_SCALAR_SETS].data_ptr(), n_loc, d_tmp.data_ptr(), d_tmp[8:].data_ptr(), stream=stream)
            torch.cuda.synchronize()
        prof_alone = lib.profile_end()

        # ---- correctness of what was timed: closed form via an independent kernel path (scalarMul)
        all_k = expect_k
        if world > 1:
            gathered = [None] * world
            dist.all_gather_object(gathered, expect_k)
            all_k = [sum(gk[s] for gk in gathered) % api.R_MOD for s in range(N_SCALAR_SETS)]
        want = [api.MSM.scalarMul(g, api.fr_from_int(k)) for k in all_k]
        res = d_res.cpu().numpy().view(np.uint64)
        for j in range(steps * per_step):
            wxy, winf = want[j % N_SCALAR_SETS]
            assert int(res[j, 8] & 0xFF) == winf and np.array_equal(res[j, :8], wxy), f"MSM result mismatch at MSM {j}"
        wd.beat(f"2^{logn}: every timed result checked")

        return {"n": n, "n_loc": n_loc, "elapsed": elapsed, "prof": prof, "prof_alone": prof_alone, "setup_s": setup_s, "bases_xy": bases_xy,
                "table_build_ms": table_build_ms, "table_bytes": bases.table_bytes(),
                "d_scalars": d_scalars, "want": want, "bases": bases}

    # ranks the collective really spans (an all-reduce of ones over the process group the partials travel on: RCCL when backend = nccl)
    collective_ranks = None
    if use_dist:
        ones = torch.ones(1, dtype=torch.int64, device=dev if dist_backend == "nccl" else "cpu")
        dist.all_reduce(ones)
        collective_ranks = {"backend": "rccl" if dist_backend == "nccl" else dist_backend, "ranks": int(ones.item())}
        if collective_ranks["ranks"] != world:  # before any timed region: a group that does not span the N ranks would time something else
            msg = f"bench.py --gpus {world}: rank {rank} sees a {collective_ranks['backend']} group of {collective_ranks['ranks']} ranks, expected {world}"
            if rank == 0:
                print(json.dumps({"error": msg, "rank": rank}), flush=True)
            raise SystemExit(msg)
    per_step = max(1, args.msms_per_step)
    m = run_size(args.logn, args.steps, args.warmup, per_step)
    n, n_loc, elapsed, prof, setup_s = m["n"], m["n_loc"], m["elapsed"], m["prof"], m["setup_s"]
    table_build_ms, table_bytes = m["table_build_ms"], m["table_bytes"]
    prof_alone = m["prof_alone"]
    bases_xy, d_scalars, want, bases = m["bases_xy"], m["d_scalars"], m["want"], m["bases"]
    m_plan = bases.plan()

    def core_object():
        """the line's contract keys, config, roofline (and the per-kernel times under extra) from the timed region alone — built twice when
        N > 1: right after the timed region (a provisional line, so that the first run on a multi-GPU node leaves its headline on stdout
        even if one of the extras behind it fails) and at the end"""
        ms_per_step = elapsed / args.steps * 1e3
        value = args.steps * per_step / elapsed
        ms_per_msm = ms_per_step / per_step
        # roofline of the dominant kernel from its duration running BY ITSELF (one stream, outside the timed region): this is the
        # figure rocprofv3 --kernel-trace --stats reports for the serial run (profiles/r2*_kernel_stats_streams1.csv). Inside the timed
        # region kernels of three streams share the GPU and a HIP-event bracket stretches beyond the per-MSM step time; that
        # overlapped figure is kept beside it for transparency only.
        acc_ms, acc_cnt = prof["msm_accumulate"]
        acc_overlapped_ms = acc_ms / max(acc_cnt, 1)
        alone_ms = prof_alone["msm_accumulate"][0] / max(prof_alone["msm_accumulate"][1], 1)
        # a long MSM runs as point slices: several accumulate launches per MSM, each over its share of the points
        SERIAL_MSMS = 6  # the serial leg above
        launches = {k: v[1] / SERIAL_MSMS for k, v in prof_alone.items()}
        acc_launches = max(launches.get("msm_accumulate", 1.0), 1.0)
        alg_bytes = 96.0 * n_loc / acc_launches  # 64 B affine point + 32 B scalar per point on this rank, per launch
        achieved = alg_bytes / (alone_ms * 1e-3) / 1e9 if alone_ms else 0.0
        out = {
            "metric": "BN254 G1 MSM/sec", "value": value, "unit": "MSM/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"msm_g1_2^{args.logn}", "points": n, "points_per_gpu": n_loc, "msms_per_step": per_step,
                       "ms_per_msm": ms_per_msm,
                       "mode": "table-less (expected_uses = 1)" if args.precompute == 1 else "precomputed table of multiples",
                       "arithmetic": "256-bit Montgomery field elements as 32-bit limbs (9x29-bit lazy limbs in the MSM), integer only",
                       "bases": "(i+1)*G resident in HBM (table of 2^(c*l)*P_i built once at upload, like an SRS)", "scalars": "uniform mod r, splitmix64 seed 0x5A4F4C54, resident in HBM",
                       "sharding": f"contiguous chunks + {dist_backend} all-gather of the step's Jacobian partials (96 bytes per MSM, one exchange per stream and step)" if world > 1 else "single GPU",
                       "collective_ranks": collective_ranks,
                       "streams": nstreams,
                       "bit_exact_check": "closed form (sum s_i*(i+1))*G via scalarMul kernel, every timed MSM"},
            "roofline": {"bound": "hbm", "kernel": "msm_accumulate_chunk_kernel", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": None, "traffic_source": TRAFFIC_SOURCE,
                         "algorithmic_bytes_per_launch": alg_bytes, "launches_per_msm": acc_launches, "avg_launch_ms": alone_ms,
                         "avg_launch_ms_source": "HIP events around the kernel, one stream in flight (6 serial MSMs right after the timed region)",
                         "avg_launch_ms_overlapped": acc_overlapped_ms,
                         "note": "MSM is integer-ALU-bound (Fp products per bucket addition x windows per point); see DESIGN.md"},
            "extra": {"kernel_ms_per_msm_overlapped": {k: (v[0] / max(v[1], 1)) * launches.get(k, 1.0) for k, v in prof.items() if v[1]},
                      "kernel_ms_per_msm_alone": {k: v[0] / SERIAL_MSMS for k, v in prof_alone.items() if v[1]},
                      "kernel_launch_sets_per_msm": acc_launches,
                      "setup_seconds": setup_s},
        }
        assert alone_ms <= ms_per_msm * 1.5 or world > 1 or nstreams == 1, "kernel-alone duration inconsistent with the step time"

        # the ceiling that actually binds msm_accumulate: VALU issue. One mixed add compiles to MADD_ISSUE_CYCLES issue cycles
        # per wave (static instruction mix of the kernel's fast path); peak = 1024 SIMDs x 2.4 GHz.
        plan_c, plan_w, plan_l = m_plan
        adds = float(n_loc) * plan_w * (1.0 - 2.0 ** -plan_c) / acc_launches  # one table row per non-zero signed c-bit digit, per launch
        issue = adds / 64.0 * MADD_ISSUE_CYCLES / (alone_ms * 1e-3) / 1e9 if alone_ms and args.logn >= 15 else None
        out["config"]["window_bits"], out["config"]["windows"], out["config"]["table_levels"] = plan_c, plan_w, plan_l
        # what the headline mode costs before its first MSM: the table of precomputed multiples (review item: "the line does not price the table")
        out["config"]["table_build_ms"], out["config"]["table_bytes"] = table_build_ms, table_bytes
        out["config"]["table_build_note"] = ("zg_g1_bases_upload_dev to completion, host-timed on this rank: plan + workspaces + msm_precompute_kernel (levels x 64 B per "
                                             "base); breakeven_msms is filled in from extra.msm_no_precompute (the same MSM with expected_uses = 1: no table)")
        out["config"]["breakeven_msms"] = None
        out["roofline"]["avg_launch_ms_alone"] = alone_ms  # same number as avg_launch_ms (kept under its round-1 name)
        out["roofline"]["valu_issue"] = {"achieved": issue, "peak": VALU_PEAK_GCYC, "unit": "G issue-cycles/s",
                                         "frac": issue / VALU_PEAK_GCYC if issue else None,
                                         "duration": "avg_launch_ms_alone (the kernel running by itself, outside the timed region)",
                                         "model": "1467 v_mad_u64_u32 + 371 other quarter-rate + 382 half-rate VALU instructions per mixed add"}
        floor_ms = adds / 64.0 / 1024.0 * MADD_MIN_NS_PER_SIMD * 1e-6 if args.logn >= 15 else None
        out["roofline"]["valu_issue_measured_rates"] = {
            "floor_ms": floor_ms, "frac": floor_ms / alone_ms if floor_ms and alone_ms else None,
            "model": "adds / (64 lanes x 1024 SIMDs) x least ns per wave-wide mixed add at the issue times measured by tools/microbench.hip "
                     "(profiles/r1_microbench_instruction_rates.txt); frac = floor / avg_launch_ms_alone"}
        default_plan = world == 1 and args.window_bits == 0 and args.precompute == 0
        pmc_all = load_pmc() if default_plan else {}
        out["roofline"]["traffic"] = measured_traffic(pmc_all, args.logn, adds)
        if out["roofline"]["traffic"] is None:
            out["roofline"]["traffic_source"] = ("null: " + ("profiles/r6_pmc.json has no counters for this size" if default_plan else
                                                            "counters were collected for the default plan on one GPU only"))
        prof_ms = profiled_duration_ms(args.logn) if default_plan else None
        if prof_ms:
            # the same kernel in the committed rocprofv3 run (profiles/r6_kernel_stats_trace1.csv: the one-stream serial bench; since round 6 a
            # launch there takes every chunk slot, like the lone launch the HIP events above bracket — msm.hip, "not company"): both fractions
            out["roofline"]["rocprofv3_avg_launch_ms"] = prof_ms
            out["roofline"]["frac_at_rocprofv3_duration"] = alg_bytes / (prof_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS
        pmc = pmc_all.get(f"2^{args.logn}", {}).get("SQ_INSTS_VALU")
        if pmc and alone_ms:
            wave_adds = adds / 64.0
            per_add = pmc / wave_adds
            ns_per_add = MADD_MIN_NS_PER_SIMD + max(per_add - MADD_STATIC_INSTRS, 0.0) * SIMPLE_OP_NS
            floor_pmc = wave_adds / 1024.0 * ns_per_add * 1e-6
            out["roofline"]["valu_issue_measured_rates"].update({
                "pmc_wave_instructions_per_add": per_add, "static_wave_instructions_per_add": MADD_STATIC_INSTRS,
                "floor_ms_from_pmc_count": floor_pmc, "frac_from_pmc_count": floor_pmc / alone_ms,
                "residual": 1.0 - floor_pmc / alone_ms,
                "residual_is": "time the SIMDs do not spend issuing: the launch runs 7/8 of the chip's chunk slots when MSMs on other streams are in flight (full "
                               "when alone), the random 64-byte table rows (DESIGN 4a: a wave waits on its gathers when the other wave of its SIMD "
                               "does too), and the chunk-length spread at the end of the launch",
                "pmc_source": "SQ_INSTS_VALU, rocprofv3 --pmc (own pass), profiles/r6_pmc.json"})
        return out, ms_per_msm

    if world > 1 and rank == 0:
        prov, _ = core_object()
        prov["provisional"] = True
        sys.stdout.flush()
        print(compact_line(prov, extra_file=None), flush=True)  # an EARLIER stdout line; the last line of a complete run supersedes it

    # BASELINE config 4: the 2^22-point MSM sharded over the ranks (same code path, untimed setup), reported beside the
    # headline workload so that the strong-scaling curve exists at both sizes of the metric
    sharded_22 = None
    if world > 1 and not args.no_extra and args.logn != 22:
        del m
        bases.free()
        m22 = run_size(22, 12, 3, 1)
        sharded_22 = {"value": 12 / m22["elapsed"], "unit": "MSM/s", "ms_per_msm": m22["elapsed"] / 12 * 1e3,
                      "points": m22["n"], "points_per_gpu": m22["n_loc"], "n_gpus": world}
        m22["bases"].free()
        bases = None
        del m22

    sharded_sc = None
    if use_dist and not args.no_extra and world & (world - 1) == 0:
        wd.beat("sharded sumcheck")
        sharded_sc = sharded_sumcheck_measurement(lib, api, torch, dist, dev, stream, world, rank, dist_backend)

    # the one-process / several-GPU C ABI on the same devices (extra, never the headline): rank 0 runs it as a child process while
    # the other ranks wait on a HOST barrier (gloo), their GPUs idle and their tables freed
    single_proc = None
    if not args.no_extra and args.logn == 20:
        if world > 1 and bases is not None:  # N = 1 keeps its handle for the extras below (288 GB: the child's table fits beside it)
            bases.free()
            bases = None
        torch.cuda.empty_cache()
        host_pg = dist.new_group(backend="gloo") if use_dist and world > 1 else None
        wd.beat("single-process child (rank 0 runs it, the others wait on a host barrier)", limit=400)
        if rank == 0:
            single_proc = single_process_child(min(world, torch.cuda.device_count()), wd=wd)  # (the gloo debugging mode shares one GPU)
        if host_pg is not None:
            dist.barrier(group=host_pg)

    wd.beat("results")
    if rank != 0:
        dist.destroy_process_group()
        return

    out, ms_per_msm = core_object()
    if single_proc is not None:
        out["extra"]["single_process_c_abi"] = single_proc
    if sharded_sc is not None:
        out["extra"]["sumcheck_v20_sharded"] = sharded_sc
    if sharded_22 is not None:
        out["extra"]["msm_2^22_sharded"] = sharded_22
    if not args.no_extra and world == 1:
        try:  # the headline above is measured and checked: nothing behind it may take the line down
            # the host-buffer entry point (what an unmodified MSM.compute call site pays): scalars cross PCIe every call
            wd.beat("extras (child processes for 2^22, the table-less plan and the compiled host loops: each restarts the clock with its own limit)", limit=300)
            h_sc = d_scalars[0].cpu().numpy().view(np.uint64)
            bases.msm(h_sc)
            t0 = time.perf_counter()
            for _ in range(3):
                hxy, hinf = bases.msm(h_sc)
            out["extra"]["msm_host_scalars_ms"] = (time.perf_counter() - t0) / 3 * 1e3
            assert hinf == want[0][1] and np.array_equal(hxy, want[0][0])
            out["extra"].update(extra_measurements(lib, api, torch, dev, stream, args, bases_xy if args.logn == 20 else None, wd=wd))
            npc = out["extra"].get("msm_no_precompute", {})
            if "ms_per_msm" in npc:
                serial_table = sum(out["extra"]["kernel_ms_per_msm_alone"].values())
                serial_plain = sum(npc.get("kernel_ms_per_msm_alone", {}).values())
                gain_p, gain_s = npc["ms_per_msm"] - ms_per_msm, serial_plain - serial_table
                out["config"]["breakeven_msms"] = {
                    "pipelined": table_build_ms / gain_p if gain_p > 0 else None, "one_at_a_time": table_build_ms / gain_s if gain_s > 0 else None,
                    "table_less_ms_per_msm": {"pipelined": npc["ms_per_msm"], "one_at_a_time_kernels": serial_plain},
                    "table_ms_per_msm": {"pipelined": ms_per_msm, "one_at_a_time_kernels": serial_table},
                    "note": "table_build_ms / (table-less ms per MSM - table ms per MSM); below this many MSMs over one SRS the table-less plan "
                            "(zg_msm_config.expected_uses = 1) is the faster choice"}
        except Exception as exc:  # noqa: BLE001
            out["extra"]["extras_error"] = repr(exc)[:300]
    if not args.no_cpu_baseline and world == 1:
        wd.beat("cpu baseline", limit=600)
        try:
            out["cpu_baseline"] = cpu_baseline(bases_xy, d_scalars[0].cpu().numpy().view(np.uint64), want[0], args.logn)
        except Exception as exc:  # noqa: BLE001  (a line without cpu_baseline is still a measured headline; the error says why it is missing)
            out["cpu_baseline_error"] = repr(exc)[:300]
    if use_dist:
        dist.destroy_process_group()
    emit(out, full_line=args.full_line)  # the LAST thing this process prints


class _Done:
    def __init__(self, rc, out, err):
        self.returncode, self.stdout, self.stderr = rc, out, err


def run_child(cmd, timeout, wd=None, env=None):
    """One child process of the extras: the watchdog's stall clock is restarted with the child's own limit (+30 s) before it starts, and the
    watchdog knows the child so that it can end it before it ends this process (a child left behind would still hold the GPU)."""
    import subprocess
    prev = wd.limit if wd is not None else None
    if wd is not None:
        wd.beat(limit=timeout + 30)
    pr = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    if wd is not None:
        wd.child = pr
    try:
        so, se = pr.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        pr.kill()
        pr.communicate()
        raise
    finally:
        if wd is not None:
            wd.child = None
            wd.beat(limit=prev)
    return _Done(pr.returncode, so, se)


def single_process_child(n_devices, wd=None):
    import subprocess
    try:
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                                 "ZG_SHARDS", "ZG_SHARD_EXCHANGE")}
        out = run_child([sys.executable, os.path.abspath(__file__), "--single-process", str(n_devices), "--logn", "20"], 240, wd=wd, env=env)
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if out.returncode != 0 or not lines:
            return {"error": (out.stderr or out.stdout)[-600:], "returncode": out.returncode}
        return json.loads(lines[-1])
    except Exception as e:  # noqa: BLE001  (an extra: never take the headline line down with it)
        return {"error": repr(e)}


def sharded_sumcheck_measurement(lib, api, torch, dist, dev, stream, world, rank, dist_backend):
    """BASELINE config 3 with the 2^20-entry table sharded over the ranks: every rank builds ITS shard of
    the eq table (shared-prefix scalar), combines it with its shards of Az/Bz/Cz, then 20 rounds of
    (local sums -> 64-byte all-gather -> host toy challenge -> local fold); the last log2(world) rounds run on the
    gathered residuals. Every rank checks final_eval == the verifier's last claim. rounds/s = 20 / max-over-ranks time."""
    v, layout = 20, lib.SC_LOW_PAIR
    n_loc = (1 << v) // world
    r = lib.field_op(lib.FR, lib.OP_TO_MONT, raw_scalars(0x45515F54, 0, v))
    tabs = [torch.from_numpy(lib.field_op(lib.FR, lib.OP_TO_MONT, raw_scalars(0x53554D43 + k, rank * n_loc, n_loc)).view(np.int64)).to(dev)
            for k in range(3)]
    d_eq = torch.empty((n_loc, 4), dtype=torch.int64, device=dev)
    d_f = torch.empty((n_loc, 4), dtype=torch.int64, device=dev)
    r_loc, sc_loc = api.sharded_eq_args(r, world, rank, layout)
    reps = 3
    times = []
    for rep in range(reps + 1):
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lib.fr_eq_table_dev(r_loc, d_eq.data_ptr(), scale=sc_loc, stream=stream)
        lib.fr_spartan_combine_dev(d_eq.data_ptr(), tabs[0].data_ptr(), tabs[1].data_ptr(), tabs[2].data_ptr(), n_loc,
                                   d_f.data_ptr(), stream=stream)
        sh = api.ShardedSumcheck(api.GpuSumcheckShardBackend(d_f, layout), world, rank)
        c0 = sh.nextRound()  # round 0 message also yields the claim g0 + g1
        claim = api._limbs((2 * api._int(c0[0]) + api._int(c0[1])) % api.R_MOD)
        ver = api.Sumcheck.Verifier(claim)
        coeffs = c0
        for rd in range(v):
            if rd:
                coeffs = sh.nextRound()
            sh.receiveChallenge(ver.verifyRound(coeffs))
        assert sh.isComplete() and np.array_equal(sh.getFinalEval(), ver.claim), "sharded sumcheck: final evaluation mismatch"
        sh.deinit()
        torch.cuda.synchronize()
        if rep:  # first repetition warms the session pool and the collective
            times.append(time.perf_counter() - t0)
    t = torch.tensor([sum(times) / len(times)], dtype=torch.float64, device=dev if dist_backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    el = float(t.item())
    return {"rounds_per_s": v / el, "ms_per_sumcheck": el * 1e3, "ranks": world, "entries_per_rank": n_loc,
            "pipeline": "per-rank eq-table shard + spartan_combine + 20 x (local sums, 64 B all-gather, host toy challenge, local fold)",
            "layout": "LOW_PAIR, contiguous shards"}


def extra_measurements(lib, api, torch, dev, stream, args, srs_xy=None, wd=None):
    """sumcheck rounds/s at v = 20 (BASELINE config 3): eq-table build + Spartan combine + 20 x (round sums,
    host toy challenge, fold) with the table resident in HBM."""
    extra = {}
    v = 20
    n = 1 << v
    r = lib.field_op(lib.FR, lib.OP_TO_MONT, raw_scalars(0x45515F54, 0, v))
    tabs = [torch.from_numpy(lib.field_op(lib.FR, lib.OP_TO_MONT, raw_scalars(0x53554D43 + k, 0, n)).view(np.int64)).to(dev)
            for k in range(3)]
    d_eq = torch.empty((n, 4), dtype=torch.int64, device=dev)
    d_f = torch.empty((n, 4), dtype=torch.int64, device=dev)
    reps = 5
    lib.profile_begin(reps * 64)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        lib.fr_eq_table_dev(r, d_eq.data_ptr(), stream=stream)
        lib.fr_spartan_combine_dev(d_eq.data_ptr(), tabs[0].data_ptr(), tabs[1].data_ptr(), tabs[2].data_ptr(), n, d_f.data_ptr(),
                                   stream=stream)
        sess = lib.SumcheckSession.open_dev(d_f.data_ptr(), n, lib.SC_HIGH_HALF, stream=stream)
        g0, g1 = sess.round_sums()
        ver = api.Sumcheck.Verifier(api._limbs((api._int(g0) + api._int(g1)) % api.R_MOD))
        while len(sess) > 1:
            g0, g1 = sess.round_sums()
            coeffs = np.stack([g0, api._limbs((api._int(g1) - api._int(g0)) % api.R_MOD)])
            sess.bind(ver.verifyRound(coeffs))
        fin = sess.final()
        assert np.array_equal(fin, ver.claim)
        sess.close()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    prof = lib.profile_end()
    extra["sumcheck_v20"] = {
        "rounds_per_s": reps * v / el, "ms_per_sumcheck": el / reps * 1e3,
        "pipeline": "eq_table + spartan_combine + 20 x (sums, host toy challenge, fold)",
        "kernel_ms": {k: (val[0] / max(val[1], 1)) for k, val in prof.items() if val[1]},
        "fold_round0_GBps": (48.0 * n) / ((prof["sc_fold"][0] / max(prof["sc_fold"][1], 1)) * 1e-3) / 1e9 if prof["sc_fold"][1] else None,
        "kernel_ms_note": "HIP-event brackets on the launch stream; eq_table's opens right behind the pageable H2D copy of r and includes "
                          "its completion — the two kernels alone take 24 us (profiles/*_sumcheck_kernel_stats.csv); sc_fold / sc_sums "
                          "are averages over the 20 rounds",
    }
    # config 3 against the HBM roof (north_star: "sumcheck-rounds/sec ... as fraction of HBM roofline"). Algorithmic bytes of the
    # protocol: the eq table written once (32 * 2^v) and every fold reading its table and writing half of it (48 * L for a table of
    # L entries, L = 2^v .. 2): 32 * 2^v + sum 48 * L. The Spartan combine in front of it (160 B per entry) is listed beside it.
    fold_bytes = 48.0 * (2 * n - 2)
    alg_bytes = 32.0 * n + fold_bytes
    ms_host = el / reps * 1e3
    kern_ms = sum(val[0] for k, val in prof.items() if k in ("eq_table", "sc_fold", "sc_sums")) / reps

    def roof(nbytes, ms, note):
        ach = nbytes / (ms * 1e-3) / 1e9 if ms else 0.0
        return {"bound": "hbm", "bytes": nbytes, "ms": ms, "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
                "note": note}
    extra["sumcheck_v20"]["roofline"] = roof(alg_bytes, ms_host, "whole protocol host-timed (eq table + 20 rounds with the verifier on the host, through the Python "
                                             "binding; the Spartan combine's 160 B per entry is not counted): latency-bound — 20 host round trips, and from round 5 on a "
                                             "round is the ~14 us floor of a launch + mailbox, not traffic")
    extra["sumcheck_v20"]["roofline_kernels_only"] = roof(alg_bytes, kern_ms, "the same bytes over the SUM of the kernels' HIP-event durations (eq table, folds, first sums)")
    extra["sumcheck_v20"]["spartan_combine_bytes"] = 160.0 * n
    # per-kernel fractions of the three HBM-shaped kernels of the path, each timed by its own HIP-event bracket on the launch stream
    per_kernel = {}
    lib.profile_begin(64)
    for _ in range(5):
        lib.fr_eq_table_dev(r, d_eq.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    pk = lib.profile_end()
    ms_eq = pk["eq_table"][0] / max(pk["eq_table"][1], 1)
    per_kernel["eq_main_kernel"] = roof(32.0 * n, ms_eq, "2^20-entry eq table, one launch (factor tables built inside): 32 B written per entry")
    lib.profile_begin(64)
    for _ in range(5):
        s2 = lib.SumcheckSession.open_spartan_dev(r, tabs[0].data_ptr(), tabs[1].data_ptr(), tabs[2].data_ptr(), lib.SC_HIGH_HALF, stream=stream)
        s2.round_sums()
        s2.close()
    pk = lib.profile_end()
    ms_sp = pk["combine"][0] / max(pk["combine"][1], 1)
    per_kernel["eq_spartan_kernel"] = roof(128.0 * n, ms_sp, "f = eq * (Az * Bz - Cz) straight into the session + round 0's sums: 96 B read + 32 B written per entry")
    s3 = lib.SumcheckSession.open_dev(d_f.data_ptr(), n, lib.SC_HIGH_HALF, stream=stream)
    s3.round_sums()
    lib.profile_begin(16)
    s3.bind(r[0])
    s3.round_sums()
    pk = lib.profile_end()
    s3.close()
    ms_f0 = pk["sc_fold"][0] / max(pk["sc_fold"][1], 1)
    per_kernel["sc_fold_kernel_round0"] = roof(48.0 * n, ms_f0, "the first fold of the 2^20-entry table (32 B read per entry, 16 B written) with the next round's sums fused")
    extra["sumcheck_v20"]["kernel_rooflines"] = per_kernel
    # runSumcheck with the toy verifier on the device as well (zg_run_sumcheck_dev): no PCIe crossing per round
    res = lib.run_sumcheck_dev(d_f.data_ptr(), n, stream=stream)
    assert res["result"] and np.array_equal(res["final_eval"], fin)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        lib.run_sumcheck_dev(d_f.data_ptr(), n, stream=stream)
    el = time.perf_counter() - t0
    extra["sumcheck_v20_device_resident"] = {"rounds_per_s": 20 * v / el, "ms_per_runSumcheck": el / 20 * 1e3,
                                             "note": "prover + toy verifier on the device; transcript equals the host-verifier run",
                                             "roofline": roof(fold_bytes + 32.0 * n, el / 20 * 1e3,
                                                              "20 folds (sum 48 * L) + the first sums pass (32 * 2^v read): 8 fold launches, then one "
                                                              "LDS-resident launch for the last 12 rounds — serial GPU code per round, not traffic")}
    # the fold by one of the reference's 128-bit challenges (stored [0, 0, lo, hi]: 9 x 5-limb product, fp29.hip.h FrMul) against a
    # full-width one, on a 2^24-entry table resident in HBM: kernel + fused next sums + mailbox, host-timed, best of 5
    try:
        n24 = 1 << 24
        d_big = d_f.repeat(n24 // n, 1)
        torch.cuda.synchronize()  # the session copies on the library's stream
        wide = lib.field_op(lib.FR, lib.OP_TO_MONT, raw_scalars(0x464F4C44, 0, 1))[0]
        nar = wide.copy()
        nar[:2] = 0
        nar[3] &= np.uint64((1 << 61) - 1)
        res = {}
        for name, c in (("full_width", wide), ("narrow", nar)):
            best = None
            for _ in range(5):
                s = lib.SumcheckSession.open_dev(d_big.data_ptr(), n24, lib.SC_LOW_PAIR)
                s.round_sums()
                t0 = time.perf_counter()
                s.bind(c)
                s.round_sums()
                dt = time.perf_counter() - t0
                s.close()
                best = dt if best is None else min(best, dt)
            res[f"{name}_us"] = best * 1e6
            res[f"{name}_TBps"] = n24 * 48 / best / 1e12
        res["note"] = "48 bytes per entry (32 read, 16 written); the transcript's challenges are the narrow kind"
        extra["fold_2^24_by_challenge_kind"] = res
        del d_big
    except Exception as exc:  # an extra: never take the headline line down with it
        extra["fold_2^24_by_challenge_kind"] = {"error": repr(exc)}
    # HyperKZG.open over the same 2^20 bases as an SRS: host table in, v quotient commitments + final value out
    if srs_xy is not None:
        try:
            params = api.HyperKZG.SetupParams(srs_xy, np.zeros(srs_xy.shape[0], dtype=np.uint8))
            res = {}
            for vv in (16, 20):
                ev = lib.field_op(lib.FR, lib.OP_TO_MONT, raw_scalars(0x4F50454E + vv, 0, 1 << vv))
                pt = lib.field_op(lib.FR, lib.OP_TO_MONT, raw_scalars(0x50543030 + vv, 0, vv))
                q0, f0 = api.HyperKZG.open(params, ev, pt, np.zeros(4, dtype=np.uint64))
                api.HyperKZG.open(params, ev, pt, np.zeros(4, dtype=np.uint64))
                ts = []
                for _ in range(5):  # median of five (the first calls size the scratch cache)
                    t0 = time.perf_counter()
                    q1, f1 = api.HyperKZG.open(params, ev, pt, np.zeros(4, dtype=np.uint64))
                    ts.append((time.perf_counter() - t0) * 1e3)
                res[f"v{vv}_ms"] = sorted(ts)[2]
                assert np.array_equal(f0, f1) and all(np.array_equal(a[0], b[0]) for a, b in zip(q0, q1))
                d_ev = torch.from_numpy(ev.view(np.int64)).to(dev)  # the same opening with the table already resident (no upload)
                lib.hyperkzg_open_dev(params._dev, d_ev.data_ptr(), 1 << vv, pt, np.zeros(4, dtype=np.uint64))
                t0 = time.perf_counter()
                for _ in range(3):
                    q2, qi2, f2 = lib.hyperkzg_open_dev(params._dev, d_ev.data_ptr(), 1 << vv, pt, np.zeros(4, dtype=np.uint64))
                res[f"v{vv}_resident_ms"] = (time.perf_counter() - t0) / 3 * 1e3
                assert np.array_equal(f2, f0) and np.array_equal(q2[0], q0[0][0])
            res["note"] = "host table in (pageable H2D included), proof out; parity with the oracle is tests/test_gpu_api_mirror.py"
            extra["hyperkzg_open"] = res
            params.deinit()
        except Exception as exc:  # an extra: never take the headline line down with it
            extra["hyperkzg_open"] = {"error": repr(exc)}
    # the reference's largest proving key (srs_size = 256 + 2^24, src/host/mod.zig:384-387): HyperKZG.setup on the device, the table's bytes and
    # one commitment with the table and without it. Parity at this size is tests/test_gpu_largest_key.py; here only the costs.
    try:
        if wd is not None:
            wd.beat(limit=300)
        nk = (1 << 24) + 256
        res = {"points": nk}
        torch.cuda.synchronize()
        d_sc = torch.from_numpy(lib.field_op(lib.FR, lib.OP_TO_MONT, raw_scalars(0x4B455924, 0, 1 << 20)).view(np.int64)).to(dev).repeat(17, 1)[:nk].contiguous()
        for uses, name in ((0, "table"), (1, "table_less")):
            t0 = time.perf_counter()
            h, _, _ = lib.Bases.hyperkzg_setup(api.generator(), api.fr_from_int(api.HyperKZG.TAU), nk, want_points=False, expected_uses=uses)
            t_setup = (time.perf_counter() - t0) * 1e3
            h.msm_dev(d_sc.data_ptr(), nk, stream=stream)
            t0 = time.perf_counter()
            for _ in range(3):
                got = h.msm_dev(d_sc.data_ptr(), nk, stream=stream)
            res[name] = {"setup_ms": t_setup, "table_bytes": h.table_bytes(), "plan": list(h.plan()), "commit_ms": (time.perf_counter() - t0) / 3 * 1e3}
            res[name + "_result"] = got
            h.free()
        a, b = res.pop("table_result"), res.pop("table_less_result")
        res["both_handles_agree"] = bool(a[1] == b[1] and np.array_equal(a[0], b[0]))
        res["note"] = "zg_hyperkzg_setup(2^24 + 256) to completion (host-timed, nothing leaves the device), then zg_msm_g1_dev over the whole key, scalars resident"
        extra["largest_key_2^24+256"] = res
        del d_sc
        torch.cuda.empty_cache()
    except Exception as exc:  # an extra: never take the headline line down with it
        extra["largest_key_2^24+256"] = {"error": repr(exc)}
    # the metric's second size, 2^22 points on this one GPU (same code path, fresh process)
    try:
        import subprocess
        out = run_child([sys.executable, os.path.abspath(__file__), "--logn", "22", "--steps", "4", "--warmup", "1", "--msms-per-step", "4",
                         "--no-cpu-baseline", "--no-extra", "--full-line", "--streams", str(args.streams)], 600, wd=wd)
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        extra["msm_2^22_single_gpu"] = {"value": d["value"], "unit": "MSM/s", "ms_per_msm": d["config"]["ms_per_msm"],
                                        "kernel_ms_per_msm_alone": d["extra"]["kernel_ms_per_msm_alone"], "roofline": d["roofline"]}
    except Exception as e:  # noqa: BLE001
        extra["msm_2^22_single_gpu"] = {"error": str(e)}
    # transparency: the same 2^20 MSM with NO table of precomputed multiples (precompute_levels = 1: classical
    # per-window bucket sets, window combine by doublings on the device) — what a one-shot caller would see
    try:
        import subprocess
        out = run_child([sys.executable, os.path.abspath(__file__), "--logn", str(args.logn), "--steps", "4", "--warmup", "1", "--msms-per-step", "8",
                         "--precompute", "1", "--no-cpu-baseline", "--no-extra", "--full-line", "--streams", str(args.streams)], 600, wd=wd)
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        extra["msm_no_precompute"] = {"value": d["value"], "unit": "MSM/s", "ms_per_msm": d["config"]["ms_per_msm"],
                                      "kernel_ms_per_msm_alone": d["extra"]["kernel_ms_per_msm_alone"]}
    except Exception as e:  # noqa: BLE001
        extra["msm_no_precompute"] = {"error": str(e)}
    # the same pipeline driven by a compiled host loop (tools/bench_sumcheck.cpp over zolt_host.hpp): what a
    # Zig/C++ prover would see, without the Python interpreter between rounds
    exe = os.path.join(ROOT, "tools", "bench_sumcheck")
    if os.path.exists(exe):
        import subprocess
        try:
            out = run_child([exe, "20", "20"], 300, wd=wd)
            extra["sumcheck_v20_compiled_host"] = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception as e:  # noqa: BLE001
            extra["sumcheck_v20_compiled_host"] = {"error": str(e)}
        # lead config 3's entry with the compiled host's figure (what a Zig / C++ prover sees); the Python-binding loop above pays the
        # interpreter and ctypes between rounds and stays beside it under its own key
        ch = extra.get("sumcheck_v20_compiled_host", {})
        if "rounds_per_s" in ch:
            py = extra["sumcheck_v20"]
            lead = {"rounds_per_s": ch["rounds_per_s"], "us_per_round": ch.get("us_per_round"), "ms_per_sumcheck": ch.get("ms_runSumcheck"),
                    "source": "compiled host loop over the C ABI (tools/bench_sumcheck.cpp, zolt_amd/host/zolt_host.hpp): 20 x (round sums from "
                              "the pinned mailbox, host toy verifier, fold launch)",
                    "device_resident": {"rounds_per_s": ch.get("device_resident_rounds_per_s"), "ms_per_runSumcheck": ch.get("device_resident_ms_runSumcheck")},
                    "roofline": roof(fold_bytes, ch["ms_runSumcheck"], "20 folds (sum 48 * L bytes) over the compiled host's whole-protocol time: "
                                     "latency-bound from round 5 on (a round is a launch + mailbox round trip)") if ch.get("ms_runSumcheck") else None,
                    "roofline_device_resident": roof(fold_bytes + 32.0 * n, ch["device_resident_ms_runSumcheck"],
                                                     "the same folds + the first sums pass, prover and verifier on the device") if ch.get("device_resident_ms_runSumcheck") else None,
                    "kernel_rooflines": py.pop("kernel_rooflines", None),
                    "python_binding": py}
            extra["sumcheck_v20"] = lead
        try:  # the sizes the reference's own runs have: prover fold sites and a Stage-2-shaped batched proof at 2^13 cycles
            out = run_child([exe, "13", "10"], 300, wd=wd)
            extra["prover_sites_v13_compiled_host"] = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception as e:  # noqa: BLE001
            extra["prover_sites_v13_compiled_host"] = {"error": str(e)}
    # ONE proof as one sequence (review item 3): proving key, three commitments from machine words, stages 1-6 with their uploads, the
    # opening — per-call costs and the top three, from compiled host code (tools/bench_prove_path.cpp); bytes at log_t = 8 are held
    # against the reference's captured proof file by tests/test_gpu_cpp_host.py
    exe = os.path.join(ROOT, "tools", "bench_prove_path")
    if os.path.exists(exe):
        import subprocess
        try:
            out = run_child([exe, "synth", "20", "2"], 600, wd=wd)
            extra["prove_path"] = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])["prove_path"]
        except Exception as e:  # noqa: BLE001
            extra["prove_path"] = {"error": str(e)}
        try:  # the same proof with a key planned for ONE use (zg_msm_config.expected_uses = 1: no table of multiples) — what a single
            # `zolt prove` run, which builds its mock SRS in-process, should ask for: three commits and an open are below the break-even
            out = run_child([exe, "synth", "20", "2", "1"], 600, wd=wd)
            one = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])["prove_path"]
            extra["prove_path_single_use_key"] = {k: one[k] for k in ("key_expected_uses", "total_ms", "total_ms_without_proving_key", "top3")}
            extra["prove_path_single_use_key"]["steps_ms"] = {st["call"].split(":")[0].split(" (")[0]: st["ms"] for st in one["steps"]
                                                              if any(w in st["call"] for w in ("proving key", "commitB", "commitM", "commitR", "HyperKZG.open"))}
        except Exception as e:  # noqa: BLE001
            extra["prove_path_single_use_key"] = {"error": str(e)}
    return extra


def cpu_baseline_2e22_full():
    """SURVEY 8(d)'s second size on the CPU, in full: bases (i+1)*G and the bench's scalars generated with the product's kernels, the
    oracle's pippengerMSM (c = 8, one thread) over all 2^22 points, the result checked against the closed form. Written to
    profiles/cpu_baseline_2e22_full.json, which cpu_baseline() then reports instead of an extrapolation."""
    import platform
    from oracle import binding as ob  # cpu_baseline leg only
    from zolt_amd import api, lib
    lib.init(0)
    n = 1 << 22
    g = api.generator()
    ks = np.zeros((n, 4), dtype=np.uint64)
    ks[:, 0] = np.arange(1, n + 1, dtype=np.uint64)
    bases_xy, _ = lib.g1_scalar_mul_batch(np.repeat(g[None, :], n, axis=0), np.zeros(n, dtype=np.uint8), lib.field_op(lib.FR, lib.OP_TO_MONT, ks))
    raw = raw_scalars(SEED, 0, n)
    sm = lib.field_op(lib.FR, lib.OP_TO_MONT, raw)
    want = api.MSM.scalarMul(g, api.fr_from_int(closed_form_scalar(raw, 0)))
    t0 = time.perf_counter()
    got, ginf = ob.msm_g1(bases_xy, None, sm)
    el = time.perf_counter() - t0
    ok = bool(ginf == want[1] and np.array_equal(got, want[0]))
    res = {"value": 1.0 / el, "unit": "MSM/s", "seconds_per_msm": el, "points": n, "cores": 1, "kind": "port", "result_checked": ok,
           "window_bits": ob.optimal_window_size(n), "host": platform.processor() or platform.machine(), "host_cores_available": os.cpu_count()}
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "cpu_baseline_2e22_full.json"), "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res))
    return 0 if ok else 1


def cpu_baseline(bases_xy, scalars, want, logn):
    """The reference's CPU path restated in C (oracle/zolt_oracle.c: pippengerMSM, c = 8, 32 windows,
    Jacobian buckets, per-window fromMontgomery) timed on this box: single thread (what HyperKZG.commit
    executes, src/poly/commitment/mod.zig:249). Cost is linear in n for n >= 32768 (fixed c = 8), so a
    bounded prefix is timed and scaled when the full size would take longer than ~30 s."""
    from oracle import binding as ob  # cpu_baseline leg only
    n = 1 << logn
    probe = min(n, 1 << 15)
    t0 = time.perf_counter()
    ob.msm_g1(bases_xy[:probe], None, scalars[:probe])
    per_point = (time.perf_counter() - t0) / probe
    sample = n
    while sample > (1 << 15) and per_point * sample > 30.0:
        sample //= 2
    t0 = time.perf_counter()
    got, ginf = ob.msm_g1(bases_xy[:sample], None, scalars[:sample])
    el = time.perf_counter() - t0
    checked = False
    if sample == n:
        assert ginf == want[1] and np.array_equal(got, want[0]), "CPU oracle disagrees with the GPU result"
        checked = True
    secs_full = el * (n / sample)
    ncores = os.cpu_count() or 1
    # the metric's other sizes (2^16 = config 1, 2^20, 2^22), same single thread, same point family: 2^16 in full
    # (also the c = 8 branch: n >= 32768); 2^22 scaled from the largest size measured (cost is linear in n at fixed c = 8)
    sizes = {}
    n16 = min(n, 1 << 16)
    t0 = time.perf_counter()
    g16, i16 = ob.msm_g1(bases_xy[:n16], None, scalars[:n16])
    el16 = time.perf_counter() - t0
    sizes["2^16"] = {"value": 1.0 / (el16 * ((1 << 16) / n16)), "unit": "MSM/s", "seconds_per_msm": el16 * ((1 << 16) / n16),
                     "sample": f"full {n16}-point MSM, single thread", "window_bits": ob.optimal_window_size(n16)}
    sizes[f"2^{logn}"] = {"value": 1.0 / secs_full, "unit": "MSM/s", "seconds_per_msm": secs_full,
                          "sample": f"{sample} of {n} points" if sample != n else "full MSM, result checked == GPU result"}
    if logn < 22:
        cached = os.path.join(ROOT, "profiles", "cpu_baseline_2e22_full.json")
        if os.path.exists(cached):  # measured in full once on an MI355X box's host (python bench.py --cpu-baseline-2e22), not re-run every time
            with open(cached) as fh:
                c22 = json.load(fh)
            sizes["2^22"] = {"value": c22["value"], "unit": "MSM/s", "seconds_per_msm": c22["seconds_per_msm"], "extrapolated": False,
                             "sample": f"full 2^22-point MSM, single thread, measured once ({c22.get('host', 'MI355X box host')}; "
                                       f"profiles/cpu_baseline_2e22_full.json), result checked == closed form: {c22.get('result_checked')}"}
        else:
            s22 = secs_full * ((1 << 22) / n)
            sizes["2^22"] = {"value": 1.0 / s22, "unit": "MSM/s", "seconds_per_msm": s22, "extrapolated": True,
                             "sample": f"EXTRAPOLATED: {sample} of {1 << 22} points, scaled linearly (c = 8 fixed for n >= 32768: 32 windows x n mixed adds)"}
    res = {"value": 1.0 / secs_full, "unit": "MSM/s", "cores": 1, "kind": "port", "sizes": sizes,
           "sample": f"{sample} of {n} points, single thread, scaled linearly to 2^{logn} (c=8 fixed for n>=32768)"
                     if sample != n else f"full 2^{logn}-point MSM, single thread, result checked == GPU result",
           "seconds_per_msm": secs_full, "oracle_lib": os.path.basename(ob.LIB_PATH), "host_cores_available": ncores,
           "result_checked": checked}
    # the reference's best case: ParallelMSM with min(cores, 8) threads (src/msm/mod.zig:673-678)
    T = min(ncores, 8)
    if T > 1:
        t0 = time.perf_counter()
        got2, ginf2 = ob.msm_g1_parallel(bases_xy[:sample], None, scalars[:sample], T)
        el2 = time.perf_counter() - t0
        res["parallel_msm"] = {"threads": T, "value": 1.0 / (el2 * (n / sample)), "unit": "MSM/s"}
    # sumcheck (config 3) on the CPU: restated runSumcheck, 20 variables, single thread
    ev = ob.f_to_mont(ob.FR, raw_scalars(0x53554D43, 0, 1 << 20))
    t0 = time.perf_counter()
    ob.run_sumcheck(ev)
    el3 = time.perf_counter() - t0
    res["sumcheck_v20"] = {"rounds_per_s": 20.0 / el3, "seconds": el3, "threads": 1}
    return res


if __name__ == "__main__":
    sys.exit(main() or 0)
