#!/usr/bin/env python3
"""Condenses a tools/profile_r1.sh output directory into the text summary committed under profiles/:
per-kernel stats (count, total/avg ns, %) from the kernel trace and per-kernel PMC averages."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("*kernel_stats.csv"):
    print("#", os.path.relpath(f, root))
    with open(f) as fh:
        for row in csv.reader(fh):
            print(", ".join(row[:8]))

agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in find("*counter_collection.csv"):
    with open(f) as fh:
        rd = csv.DictReader(fh)
        for row in rd:
            k = row.get("Kernel_Name", "?").split("(")[0]
            c = row.get("Counter_Name", "?")
            try:
                v = float(row.get("Counter_Value", "0"))
            except ValueError:
                continue
            a = agg[k][c]
            a[0] += v
            a[1] += 1
print("\n== PMC averages per dispatch (rocprofv3 --pmc, separate passes) ==")
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        tot, n = agg[k][c]
        print(f"    {c:24s} avg {tot / n:16.1f}   dispatches {n}")
