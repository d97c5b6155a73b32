#!/usr/bin/env python3
"""Timeline of the LAST burst of kernels in a rocprofv3 results .db (bursts are separated by idle gaps > 1 ms): per kernel its queue /
stream, start and end relative to the burst's first kernel (us), duration."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = next((c for c in ("queue_id", "stream_id", "queue") if c in cols), None)
rows = list(db.execute("select name, start, end" + (", " + qcol if qcol else "") + " from kernels order by start"))
bursts, cur = [], []
for r in rows:
    if cur and r[1] - max(x[2] for x in cur) > 1_000_000:
        bursts.append(cur); cur = []
    cur.append(r)
bursts.append(cur)
which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
b = bursts[which]
t0 = b[0][1]
print("bursts:", len(bursts), "kernels in this one:", len(b), "span %.1f us" % ((max(x[2] for x in b) - t0) / 1e3))
for r in b:
    print("%-6s %9.1f %9.1f %8.1f  %s" % (r[3] if qcol else "-", (r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[0].split("(")[0].replace("void ", "").replace("zg::", "")[:60]))
