#!/bin/bash
# build an A/B variant of libzolt_gpu.so into build_ab/ (git-ignored; travels with gpurun):
#   tools/build_variant.sh NAME "-DZG_FOO=1 ..."      then run with ZOLT_GPU_LIB=build_ab/libzolt_gpu_NAME.so
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; FLAGS=$2
D=$ROOT/build_ab/$NAME
mkdir -p $D
for f in runtime msm poly; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -Wall -Wno-unused-function $FLAGS \
    -I$ROOT/zolt_amd/csrc -c $ROOT/zolt_amd/csrc/$f.hip -o $D/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/build_ab/libzolt_gpu_$NAME.so $D/runtime.o $D/msm.o $D/poly.o
rm -rf $D
echo built build_ab/libzolt_gpu_$NAME.so
