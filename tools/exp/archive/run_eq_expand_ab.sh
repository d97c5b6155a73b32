#!/bin/bash
# round 5: the expanding eq-table kernel (128-bit challenges in the middle of the index) against the one-product-per-entry kernel
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for ex in 0 1; do
  for v in 14 16 20 24; do
    echo "== ZG_EQ_EXPAND=$ex v=$v narrow"; ZG_EQ_EXPAND=$ex python3 tools/bench_eq.py --v $v --reps 300 --narrow
  done
done
echo "== wide challenges (expansion never applies)"; python3 tools/bench_eq.py --v 20 --reps 300
