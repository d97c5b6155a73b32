#!/usr/bin/env python3
"""Is the 2^20 fold slower 'in the protocol' (launched on an idle GPU after a host round trip) than back to back? Four sessions on four
tables: round sums of each (host waits), then the four binds enqueued without a wait in between, then the four next sums.
  rocprofv3 --kernel-trace -d out -o cold -- python3 tools/exp/fold_cold.py ; tools/exp/dbseq.py out/.../cold_results.db sc_fold"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from zolt_amd import lib
v = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << v
lib.init(0)
rng = np.random.default_rng(3)
def rand_fr(k):
    a = rng.integers(0, 1 << 63, size=(k, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)
    return a
tabs = [lib.DeviceBuffer.from_host(rand_fr(n)) for _ in range(4)]
ch = rand_fr(1)[0]
for rep in range(6):
    ss = [lib.SumcheckSession.open_dev(t.ptr, n, lib.SC_HIGH_HALF) for t in tabs]
    for s in ss:
        s.round_sums()
    lib.sync()
    time.sleep(0.002)  # an idle GPU, as after a host round trip
    for s in ss:       # four folds back to back: the first starts cold
        s.bind(ch)
    for s in ss:
        s.round_sums()
    time.sleep(0.002)
    for s in ss:       # one fold at a time, the host waiting for each round's sums in between (the protocol's rhythm)
        s.bind(ch)
        s.round_sums()
    for s in ss:
        s.close()
print("ok")
