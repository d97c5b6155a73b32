#!/usr/bin/env python3
"""Per-kernel durations (us) of a rocprofv3 results .db, in order of first dispatch: name, calls, avg, min."""
import sqlite3, sys
for f in sys.argv[1:]:
    db = sqlite3.connect(f)
    print("==", f)
    for r in db.execute("select name, count(*), avg(end-start), min(end-start) from kernels group by name order by min(start)"):
        n = r[0].split("(")[0].replace("void ", "")
        print("%-60s %5d %9.2f %9.2f" % (n[:60], r[1], r[2] / 1e3, r[3] / 1e3))
