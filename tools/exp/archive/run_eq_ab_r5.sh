#!/bin/bash
# round 5 A/B of the eq-table / fused Spartan-open kernels: the committed build against build_ab/libzolt_gpu_old.so (the tree before the
# scratch-free factor prologue, the unpacked hi rows and the two-products-per-trip main loop)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for lib in build_ab/libzolt_gpu_old.so zolt_amd/libzolt_gpu.so; do
  for v in 16 20 24; do
    echo "== $lib v=$v"; ZOLT_GPU_LIB=$PWD/$lib python3 tools/bench_eq.py --v $v --reps 300
  done
done
