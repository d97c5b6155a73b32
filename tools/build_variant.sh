#!/bin/bash
# build an A/B variant of libzolt_gpu.so into build_ab/ (git-ignored; travels with gpurun):
#   tools/build_variant.sh NAME "-DZG_FOO=1 ..." [files to recompile, default: all]
# then run with ZOLT_GPU_LIB=build_ab/libzolt_gpu_NAME.so. Sources not named take the objects of the regular build
# (zolt_amd/csrc/*.o: run make there first).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; FLAGS=$2; FILES=${3:-"runtime msm poly psc sharded rrw rwc ingest"}
ALL="runtime msm poly psc sharded rrw rwc ingest"
D=$ROOT/build_ab/$NAME
mkdir -p $D
for f in $FILES; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -Wall -Wno-unused-function $FLAGS \
    -I$ROOT/zolt_amd/csrc -c $ROOT/zolt_amd/csrc/$f.hip -o $D/$f.o &
done
wait
OBJS=""
for f in $ALL; do
  if [ -f $D/$f.o ]; then OBJS="$OBJS $D/$f.o"; else OBJS="$OBJS $ROOT/zolt_amd/csrc/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/build_ab/libzolt_gpu_$NAME.so $OBJS -ldl -lpthread
rm -rf $D
echo built build_ab/libzolt_gpu_$NAME.so
