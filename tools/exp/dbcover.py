#!/usr/bin/env python3
"""For the kernels of a rocprofv3 results .db whose name contains PATTERN (default: accumulate): how much of the busy span has at least one
of them resident (union / span), the sum of their durations over the span (mean concurrency), and the same for everything else.
Idle gaps above 2 ms split the run; the longest piece is analysed (a pipelined timed region)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else "accumulate"
rows = list(db.execute("select name, start, end from kernels order by start"))
pieces, cur = [], []
for r in rows:
    if cur and r[1] - max(x[2] for x in cur) > 2_000_000:
        pieces.append(cur); cur = []
    cur.append(r)
pieces.append(cur)
p = max(pieces, key=lambda x: max(y[2] for y in x) - x[0][1])
t0, t1 = p[0][1], max(y[2] for y in p)
if any(pat in r[0] for r in p):  # the window from the first to the last matching kernel (set-up and checks lie outside)
    t0 = min(r[1] for r in p if pat in r[0])
    t1 = max(r[2] for r in p if pat in r[0])
    p = [r for r in p if r[2] > t0 and r[1] < t1]
def union(iv):
    iv = sorted(iv); tot = 0; ce = None; cs = None
    for s, e in iv:
        if ce is None or s > ce:
            if ce is not None: tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    if ce is not None: tot += ce - cs
    return tot
acc = [(r[1], r[2]) for r in p if pat in r[0]]
oth = [(r[1], r[2]) for r in p if pat not in r[0]]
span = t1 - t0
print("span %.3f ms, %d kernels (%d '%s')" % (span / 1e6, len(p), len(acc), pat))
print("'%s': resident %.1f %% of the span, mean concurrency %.2f, mean duration %.1f us" % (pat, 100 * union(acc) / span, sum(e - s for s, e in acc) / span, sum(e - s for s, e in acc) / max(1, len(acc)) / 1e3))
print("others: resident %.1f %% of the span, mean concurrency %.2f" % (100 * union(oth) / span, sum(e - s for s, e in oth) / span))
only_oth = span - union(acc)
print("time with NO '%s' kernel resident: %.3f ms (%.1f %%)" % (pat, only_oth / 1e6, 100 * only_oth / span))
