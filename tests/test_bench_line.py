"""The line the driver parses (round-5 review: bench.py's 22 KB single line outgrew the driver's capture window and the round went
unmeasured). CPU only: bench.compact_line is fed the full object of a real run (profiles/r5z_bench.json, the very line that was lost)
and a two-rank object; the line must stay under 4 KB, round-trip through json.loads, carry roofline / cpu_baseline and the north_star's
other figures as scalars, and be the LAST line emit() prints, with everything else in the side file.
Match: the reference's harness prints one short result per size (src/bench.zig:243-287)."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_module():
    spec = importlib.util.spec_from_file_location("bench_module_line", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def _round5_object():
    with open(os.path.join(ROOT, "profiles", "r5z_bench.json")) as fh:
        return json.loads(fh.read().strip().splitlines()[-1])


def _scalars_only(d, depth=0):
    for k, v in d.items():
        if isinstance(v, dict):
            assert depth < 1, f"nested object under {k}"
            _scalars_only(v, depth + 1)
        elif isinstance(v, list):
            assert all(not isinstance(x, (dict, list)) for x in v) and len(v) <= 2
        elif isinstance(v, str):
            assert len(v) <= 80, (k, len(v))


def test_compact_line_of_the_lost_round5_object():
    bench, full = _bench_module(), _round5_object()
    assert len(json.dumps(full)) > 20000  # the object that did not fit
    text = bench.compact_line(full)
    assert len(text) < 4096 and "\n" not in text
    line = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert k in line
    assert line["value"] == float(f"{full['value']:.6g}") and line["ms_per_step"] == float(f"{full['ms_per_step']:.6g}")
    assert line["steps"] == full["steps"] and line["warmup"] == full["warmup"] and line["n_gpus"] == 1
    cfg, roof, cpu, also = line["config"], line["roofline"], line["cpu_baseline"], line["also"]
    assert cfg["workload"] == "msm_g1_2^20" and cfg["points"] == 1 << 20 and cfg["msms_per_step"] == 32
    assert (cfg["window_bits"], cfg["windows"], cfg["table_levels"]) == (17, 15, 15)
    assert cfg["table_build_ms"] > 0 and cfg["table_bytes"] == full["config"]["table_bytes"]
    assert len(cfg["breakeven_msms"]) == 2 and all(x > 1 for x in cfg["breakeven_msms"])
    assert roof["bound"] == "hbm" and roof["kernel"] == "msm_accumulate_chunk_kernel" and roof["peak"] == 8000.0 and roof["unit"] == "GB/s"
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-6
    assert roof["traffic"] > roof["algorithmic_bytes_per_launch"] and roof["avg_launch_ms"] > 0 and roof["rocprofv3_avg_launch_ms"] > 0
    assert 0 < roof["valu_issue_frac"] <= 1.0 and roof["counters"].startswith("profiles/")
    assert cpu["kind"] == "port" and cpu["cores"] == 1 and cpu["unit"] == "MSM/s" and cpu["seconds_per_msm"] > 1 and cpu["sample"]
    # north_star: the 2^22 figure, the table-less figure and sumcheck rounds/s with its fraction, as scalars on the line
    for k in ("msm_2e22_per_s", "msm_2e22_roofline_frac", "msm_table_less_per_s", "sumcheck_rounds_per_s", "sumcheck_roofline_frac",
              "sumcheck_device_resident_rounds_per_s"):
        assert isinstance(also[k], float) and also[k] > 0, k
    _scalars_only(line)
    assert line["extra_file"] == "bench_extra.json" and "extra" not in line


def test_compact_line_two_ranks_and_missing_legs():
    """N = 2: no cpu_baseline (rank 0 at N = 1 only), no PMC traffic, sharded extras; long strings anywhere in the full object never
    reach the line."""
    bench, full = _bench_module(), _round5_object()
    full["n_gpus"] = 2
    full["config"]["points_per_gpu"] = 1 << 19
    full["config"]["collective_ranks"] = {"backend": "rccl", "ranks": 2}
    full["config"]["sharding"] = "x" * 5000
    full["roofline"]["traffic"] = None
    full["roofline"]["traffic_source"] = "y" * 5000
    del full["cpu_baseline"]
    full["extra"] = {"msm_2^22_sharded": {"value": 321.5, "note": "z" * 9000}, "sumcheck_v20_sharded": {"rounds_per_s": 12345.6},
                     "single_process_c_abi": {"error": "e" * 9000}}
    text = bench.compact_line(full)
    line = json.loads(text)
    assert len(text) < 4096 and "cpu_baseline" not in line
    assert line["config"]["collective_ranks"] == {"backend": "rccl", "ranks": 2} and line["config"]["points_per_gpu"] == 1 << 19
    assert line["roofline"]["traffic"] is None and line["roofline"]["counters"] is None
    assert line["also"] == {"msm_2e22_sharded_per_s": 321.5, "sumcheck_sharded_rounds_per_s": 12345.6, "sumcheck_host": "python binding"}
    assert "provisional" not in line
    full["provisional"] = True  # what rank 0 prints right after the timed region when N > 1 (an earlier stdout line)
    prov = json.loads(bench.compact_line(full, extra_file=None))
    assert prov["provisional"] is True and prov["extra_file"] is None and prov["value"] == line["value"]


def test_compact_line_never_outgrows_the_limit_whatever_the_strings():
    """the last line of defence: long strings in the very keys the line carries are dropped, the numbers stay"""
    bench, full = _bench_module(), _round5_object()
    full["config"]["workload"] = "w" * 5000
    full["cpu_baseline"]["sample"] = "s" * 5000
    full["roofline"]["kernel"] = "k" * 5000
    text = bench.compact_line(full)
    line = json.loads(text)
    assert len(text) < 4096 and line["value"] > 0 and line["roofline"]["frac"] > 0 and line["cpu_baseline"]["value"] > 0
    assert "workload" not in line["config"] and line["config"]["points"] == 1 << 20


def test_emit_prints_the_compact_line_last_and_writes_the_side_file(tmp_path, monkeypatch, capsys):
    bench, full = _bench_module(), _round5_object()
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))  # the profiles/ copy goes nowhere here (no such directory): only the cwd file
    print("an earlier stdout line")
    bench.emit(full)
    lines = capsys.readouterr().out.rstrip("\n").splitlines()
    assert len(lines) == 2 and len(lines[-1]) < 4096
    assert json.loads(lines[-1])["metric"] == "BN254 G1 MSM/sec"
    side = json.load(open(tmp_path / "bench_extra.json"))
    assert side == full and "prove_path" in side["extra"] and os.listdir(tmp_path) == ["bench_extra.json"]
    bench.emit(full, full_line=True)  # the extras' child processes: the full object, one line, no side file rewritten
    out = capsys.readouterr().out.strip()
    assert json.loads(out) == full
