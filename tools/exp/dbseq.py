#!/usr/bin/env python3
"""Every dispatch of the kernels whose name contains the pattern, in start order: start offset to the previous end (us), duration (us), grid."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
gx = "grid_size_x" if "grid_size_x" in cols else ("grid_x" if "grid_x" in cols else None)
prev_end = None
for r in db.execute("select name, start, end" + (", " + gx if gx else "") + " from kernels order by start"):
    gap = (r[1] - prev_end) / 1e3 if prev_end is not None else 0.0
    prev_end = r[2]
    if pat in r[0]:
        print("%-44s gap %9.2f us  dur %8.2f us %s" % (r[0].split("(")[0].replace("void ", "")[:44], gap, (r[2] - r[1]) / 1e3, r[3] if gx else ""))
