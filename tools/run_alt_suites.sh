#!/bin/bash
# The GPU suite under SETS of alternate code-path switches (docs/design/08_switches.md): every switch exists so that a path can be forced,
# and a path that is only ever forced inside the one test written for it meets few of the library's entry points. Round 6 found a GPU
# memory fault this way (HyperKZG.open's fused long levels with ZG_MSM_TWO_PASS_SORT=0). Run on the GPU box from the repo root:
#     bash tools/run_alt_suites.sh [set ...]        # default: all sets; logs in gpurun_out/alt_suites/, one summary line per set
# A test that ASSERTS a default (a window plan, a launch shape) may fail under a set that overrides it: read the failure, it is not
# necessarily a bug. profiles/r6_alternate_switch_suites.txt is the round-6 run.
OUT=gpurun_out/alt_suites
mkdir -p $OUT
declare -A SETS=(
  [alt1]="ZG_MSM_ALONE_FULL=0 ZG_MSM_TWO_PASS_SORT=0 ZG_MSM_REDUCE_2D=0"
  [alt2]="ZG_DEV_ALLOC_CACHE_MB=0 ZG_MSM_HOST_AFFINE=0 ZG_MSM_SIDE_TABLE=0"
  [alt3]="ZG_MSM_LDS_SORT=0 ZG_MSM_BATCH_FUSE=0 ZG_MSM_ROWS_SHARED_TAIL=0"
  [alt4]="ZG_MSM_TABLE_SPAN_MB=64 ZG_MSM_LANES=2 ZG_MSM_HOST_SLICES=2"
  [alt5]="ZG_SC_MAX_BLOCKS=1 ZG_SC_TAIL_MAX=1"
  [alt6]="ZG_SC_FOLD_THREADS=64 ZG_SC_SUMS_THREADS=64 ZG_EQ_BLOCKS=1024 ZG_EQ_WG=4 ZG_EQ_EXPAND=14 ZG_SC_SPARTAN_WG=1"
  [alt7]="ZG_PSC_BLOCKS=1 ZG_PSC_POOL=0 ZG_PSC_SPREAD_MAX_PAIRS=0 ZG_RRW_MASKED_FOLDS=0"
  [alt8]="ZG_MSM_CHUNK_SCHED=0 ZG_MSM_PRECOMPUTE=2"
  [alt9]="ZG_HK_FUSE_LONG=0 ZG_MSM_PRECOMPUTE_V1=1 ZG_MSM_ROWCOL_WAVE_FROM=1 ZG_FB_WINDOW_BITS=8"
  [alt10]="ZG_ROWS_SMALL_COEFF=0 ZG_ROWS_STAGE=0 ZOLT_WITNESS_SLICES=1 ZOLT_HOST_THREADS=1"
)
NAMES=${@:-alt1 alt2 alt3 alt4 alt5 alt6 alt7 alt8 alt9 alt10}
for name in $NAMES; do
  env ${SETS[$name]} timeout 1200 python -m pytest tests -m gpu -q > $OUT/$name.log 2>&1
  echo "$name rc=$? [${SETS[$name]}]: $(tail -1 $OUT/$name.log | cut -c1-160)"
  grep -a "^FAILED\|Fatal" $OUT/$name.log | head -10 | cut -c1-240
done
