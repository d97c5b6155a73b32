#!/bin/bash
# Round 4: rocprofv3 evidence for the sumcheck-family kernels at ONE size (2^20 entries) — durations, then the counters in passes of
# their own (never combined with a tracing domain): bash tools/profile_r4_sc.sh <tag> [log2 size]
TAG=${1:-r4}
LOGN=${2:-20}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_sc_$TAG
export TMPDIR=/tmp
mkdir -p $OUT
cd /tmp
CMD="python3 $ROOT/tools/prof_sc_kernels.py $LOGN 8"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD </dev/null > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD </dev/null > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD </dev/null > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_sq -- $CMD </dev/null > $OUT/pmc_sq.log 2>&1
{
  echo "# command: $CMD   (every kernel below ran on 2^$LOGN-entry tables only; FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them:"
  echo "#  FETCH_SIZE counts a wide coalesced streaming read at half its bytes on gfx950 — MI355X_MICROARCH.md — double it before comparing)"
  python3 $ROOT/tools/summarize_prof.py $OUT
} > $OUT/summary.txt 2>&1
find $OUT -name "*.csv" -size +2M -delete
tail -5 $OUT/trace.log
