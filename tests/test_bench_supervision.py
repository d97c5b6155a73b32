"""bench.py --gpus N supervision on the CPU (no GPU is touched: the injected faults fire before anything imports torch.cuda): a rank that
stalls or dies takes the whole job down with a non-zero exit code, a JSON line that names the rank, and per-rank logs on disk — the
launcher kills exactly the fresh child processes it started. (Round-4 review item 7; the N > 1 data path itself is covered by
tests/test_sharded_gloo.py and tests/test_gpu_multidev.py.)"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(fault, tmp_path, stall_s="3"):
    env = dict(os.environ, ZOLT_BENCH_FAULT=fault, ZOLT_BENCH_STALL_S=stall_s, ZOLT_BENCH_LOG_DIR=str(tmp_path))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.time()
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=120)
    return res, time.time() - t0


def test_a_stalled_rank_stops_the_job(tmp_path):
    res, el = run("stall:1", tmp_path)
    assert res.returncode != 0 and el < 60
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert line["rank"] == 1 and "rank 1 failed" in line["error"] and "no progress" in line["stderr_tail"]
    log = open(os.path.join(tmp_path, "rank1.log")).read()
    assert "injected stall" in log and "giving up" in log and "Thread" in log  # phase, verdict, tracebacks of every thread
    assert os.path.exists(os.path.join(tmp_path, "rank0.stderr"))


def test_a_rank_that_exits_stops_the_job(tmp_path):
    res, el = run("exit:0", tmp_path, stall_s="60")
    assert res.returncode != 0 and el < 30  # the healthy rank is killed by the launcher, not waited for
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert line["rank"] == 0 and "exit code 7" in line["error"]
