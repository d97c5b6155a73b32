#!/usr/bin/env python3
"""profiles/<tag>_pmc.json from a tools/collect_profiles.sh output directory: per launch of the dominant kernel (msm_accumulate_chunk_kernel)
the memory-side counters, the executed VALU instructions, the stall split and the average duration, at 2^20 and 2^22 points. bench.py
reads THIS file for roofline.traffic and roofline.valu_issue_measured_rates (no hand-typed constants; tests/test_abi_and_host.py holds
bench.py and the file together).  python tools/pmc_json.py <tag> <dir 2^20> <dir 2^22>"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

KERNEL = "msm_accumulate_chunk_kernel"


def counters(root):
    agg = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if KERNEL not in row.get("Kernel_Name", ""):
                    continue
                try:
                    v = float(row.get("Counter_Value", "0"))
                except ValueError:
                    continue
                a = agg[row.get("Counter_Name", "?")]
                a[0] += v
                a[1] += 1
    return {k: {"avg": t / n, "dispatches": n} for k, (t, n) in agg.items() if n}


def duration(root):
    for f in glob.glob(os.path.join(root, "trace1", "**", "*kernel_stats.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if KERNEL in row.get("Name", ""):
                    return {"calls": int(row["Calls"]), "avg_us": float(row["AverageNs"]) / 1e3, "total_ms": float(row["TotalDurationNs"]) / 1e6}
    return None


def bench_line(root):
    try:
        lines = [l for l in open(root.rstrip("/") + "/trace1.log") if l.startswith("{")]
        d = json.loads(lines[-1])
        return {"value": d["value"], "ms_per_msm": d["config"]["ms_per_msm"], "window_bits": d["config"].get("window_bits"), "windows": d["config"].get("windows"),
                "launches_per_msm": d["roofline"].get("launches_per_msm")}
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)}


def calibration(root):
    out = {}
    for mode in ("gather", "stream"):
        try:
            rows = [l for l in open(os.path.join(root, f"cal_{mode}.log")) if "rows of 64 B" in l]
            mb = float(re.search(r"= ([\d.]+) MB per launch", rows[-1]).group(1))
            c = [0.0, 0]
            for f in glob.glob(os.path.join(root, f"cal_{mode}", "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") == "FETCH_SIZE" and f"k_{mode}64" in row.get("Kernel_Name", ""):  # not the buffer fills in front
                        c[0] += float(row["Counter_Value"]); c[1] += 1
            out[mode] = {"bytes_per_launch_MB": mb, "FETCH_SIZE_KB_avg": c[0] / c[1] if c[1] else None,
                         "reported_over_actual": (c[0] / c[1] * 1024 / (mb * 1e6)) if c[1] else None}
        except Exception as e:  # noqa: BLE001
            out[mode] = {"error": repr(e)}
    return out


def main():
    tag, d20, d22 = sys.argv[1], sys.argv[2], sys.argv[3]
    res = {"tag": tag, "kernel": KERNEL, "generated_by": "tools/collect_profiles.sh -> tools/pmc_json.py",
           "units": {"FETCH_SIZE": "KB as rocprofv3 reports it (gfx950: a wide coalesced stream is tallied at half its bytes; random 64-byte rows at face value)",
                     "WRITE_SIZE": "KB", "SQ_*": "per launch; WAVE_CYCLES / WAIT_* / ACTIVE_INST_* in quad-cycles"},
           "sizes": {}}
    for name, d in (("2^20", d20), ("2^22", d22)):
        res["sizes"][name] = {"counters": counters(d), "duration": duration(d), "bench": bench_line(d)}
    res["calibration"] = calibration(d20)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
