#!/bin/bash
# accumulate chunk slots with other MSMs in flight: 131072 = every slot, 122880 = 15/16 (the default), others for the sweep
mkdir -p gpurun_out/nt_sweep
for NT in ${NTS:-default 131072 126976 122880 118784 114688}; do
  if [ $NT = default ]; then E=""; else E="ZG_MSM_CHUNK_THREADS=$NT"; fi
  for rep in 1 2; do
    v=$(env $E python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extra </dev/null 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f MSM/s' % d['value'])")
    echo "NT=$NT rep=$rep: $v"
  done
done | tee gpurun_out/nt_sweep/sweep.txt
