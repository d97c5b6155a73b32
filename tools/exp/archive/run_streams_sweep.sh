#!/bin/bash
mkdir -p gpurun_out/nt_sweep
for ST in 3 4 5 6 3; do
  for Q in "" 8; do
    if [ -n "$Q" ]; then export GPU_MAX_HW_QUEUES=$Q; else unset GPU_MAX_HW_QUEUES; fi
    v=$(python bench.py --steps 20 --warmup 4 --streams $ST --no-cpu-baseline --no-extra </dev/null 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f MSM/s' % d['value'])")
    echo "streams=$ST hw_queues=${Q:-default}: $v"
  done
done | tee gpurun_out/nt_sweep/streams.txt
