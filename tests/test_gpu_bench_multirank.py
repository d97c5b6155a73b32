"""bench.py's N > 1 code on the one-GPU test box: two ranks over the gloo rendezvous share cuda:0 (RCCL refuses two ranks per
device), so every branch the driver's `--gpus N` run takes — the self-launcher, run_size at 2^20 and 2^22 with sharded bases, the
sharded sumcheck measurement, the one-process / several-GPU child, the host barrier, the max-over-ranks timing and the single JSON
line of rank 0 — executes here before it ever meets an 8-GPU node. Numbers from this run are never reported."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, extra_env, cwd, timeout=900):
    """runs bench.py in `cwd`; returns (the compact line the driver parses, the side file rank 0 wrote there)"""
    env = dict(os.environ, **extra_env)
    env.pop("ZG_SHARDS", None)
    env.pop("ZG_SHARD_EXCHANGE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=str(cwd))
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    n_ranks = int(args[args.index("--gpus") + 1]) if extra_env.get("ZOLT_BENCH_DIST_BACKEND") else 1
    # N > 1: a provisional line right after the timed region (the headline survives a failing extra on a first multi-GPU run), then the line
    assert out.returncode == 0 and len(lines) == (2 if n_ranks > 1 else 1), (out.returncode, out.stdout[-800:], out.stderr[-1500:])
    if n_ranks > 1:
        prov = json.loads(lines[0])
        assert prov["provisional"] is True and prov["n_gpus"] == n_ranks and prov["value"] > 0 and "roofline" in prov and len(lines[0]) < 4096
    last = out.stdout.rstrip("\n").splitlines()[-1]
    assert last == lines[-1] and len(last) < 4096 and "provisional" not in json.loads(last), (len(last), out.stdout[-600:])  # the compact line is the LAST line of stdout
    side = os.path.join(str(cwd), "bench_extra.json")
    assert os.path.exists(side), "rank 0 writes the side file into the cwd"
    assert sorted(os.listdir(str(cwd))) == ["bench_extra.json"], "one side file, written by rank 0 only"
    return json.loads(last), json.load(open(side))


def test_bench_two_ranks_over_gloo_takes_every_multi_rank_branch(tmp_path):
    line, d = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--msms-per-step", "2", "--no-cpu-baseline"], {"ZOLT_BENCH_DIST_BACKEND": "gloo"}, tmp_path)
    assert line["n_gpus"] == 2 and line["value"] == pytest.approx(d["value"], rel=1e-5) and line["extra_file"] == "bench_extra.json"
    assert line["config"]["collective_ranks"] == {"backend": "gloo", "ranks": 2} and "roofline" in line
    assert line["also"]["msm_2e22_sharded_per_s"] > 0 and line["also"]["sumcheck_sharded_rounds_per_s"] > 0
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "strong"
    assert d["metric"] == "BN254 G1 MSM/sec" and d["unit"] == "MSM/s" and d["value"] > 0
    assert d["config"]["points"] == 1 << 20 and d["config"]["points_per_gpu"] == 1 << 19
    assert d["config"]["collective_ranks"] == {"backend": "gloo", "ranks": 2}  # the all-reduce census of the group the partials travel on
    ex = d["extra"]
    assert ex["msm_2^22_sharded"]["points_per_gpu"] == 1 << 21 and ex["msm_2^22_sharded"]["value"] > 0
    assert ex["sumcheck_v20_sharded"]["ranks"] == 2 and ex["sumcheck_v20_sharded"]["rounds_per_s"] > 0
    sp = ex["single_process_c_abi"]
    assert "error" not in sp, sp
    assert sp["devices"] >= 1 and any(isinstance(v, dict) and v.get("msm_per_s", 0) > 0 for v in sp.values())


def test_bench_single_rank_through_the_sharded_path(tmp_path):
    """world size 1 with the whole sharded sequence forced (partial MSM, RCCL all-gather with its one rank, device combine)"""
    env = {"ZOLT_BENCH_FORCE_SHARDED": "1", "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29731"}
    line, d = _bench(["--gpus", "1", "--steps", "2", "--warmup", "1", "--msms-per-step", "2", "--no-cpu-baseline", "--no-extra"], env, tmp_path)
    assert d["n_gpus"] == 1 and d["value"] > 0 and line["n_gpus"] == 1 and line["config"]["collective_ranks"]["ranks"] == 1
