#!/usr/bin/env python3
"""Timeline of a rocprofv3 --kernel-trace CSV of pipelined MSMs: per stream (queue) kernel start/end relative to the first
accumulate launch of the steady state, and how much of the wall time has an accumulate kernel resident.
usage: trace_timeline.py <kernel_trace.csv> [first_acc_index] [n_acc]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
k0 = int(sys.argv[2]) if len(sys.argv) > 2 else 12
nk = int(sys.argv[3]) if len(sys.argv) > 3 else 6
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("zg::", "").replace("void ", ""), r["Queue_Id"]) for r in rows]
ev.sort()
acc = [e for e in ev if e[2].startswith("msm_accumulate_chunk")]
# skip the MSMs of the setup / warmup: take the run of accumulate launches k0 .. k0+nk
t0, t1 = acc[k0][0], acc[k0 + nk][0]
print(f"window: {nk} MSMs in {(t1 - t0) / 1e6:.3f} ms -> {(t1 - t0) / 1e6 / nk:.3f} ms/MSM")
queues = sorted({e[3] for e in ev if t0 <= e[0] < t1})
for e in ev:
    if t0 <= e[0] < t1:
        col = queues.index(e[3])
        print(f"{(e[0] - t0) / 1e3:9.1f} {(e[1] - t0) / 1e3:9.1f} {(e[1] - e[0]) / 1e3:8.1f} us  " + "    " * col + f"q{col} {e[2][:44]}")
# union coverage of accumulate kernels
iv = sorted((max(a, t0), min(b, t1)) for a, b, n, q in acc if b > t0 and a < t1)
cov, cur_a, cur_b = 0, None, None
for a, b in iv:
    if cur_b is None or a > cur_b:
        if cur_b is not None: cov += cur_b - cur_a
        cur_a, cur_b = a, b
    else:
        cur_b = max(cur_b, b)
if cur_b is not None: cov += cur_b - cur_a
print(f"accumulate resident {100.0 * cov / (t1 - t0):.1f} % of the window; mean accumulate duration {sum(b - a for a, b in iv) / len(iv) / 1e3:.1f} us")
