#!/bin/bash
# rocprofv3 recipe for the round's committed profiles (run on the GPU box from the repo root):
#   bash tools/profile_r1.sh <tag> [bench args]
# pass 1: kernel trace + stats; passes 2..4: PMC counters alone (never combined with tracing domains).
TAG=${1:-r1}
shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
export TMPDIR=/tmp
mkdir -p $OUT
cd /tmp
BENCH="python3 $ROOT/bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extra $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH </dev/null > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH </dev/null > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH </dev/null > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/pmc_sq -- $BENCH </dev/null > $OUT/pmc_sq.log 2>&1
cd $OUT
echo "# bench command: $BENCH" > $OUT/summary.txt
grep -h '^{' $OUT/trace.log | tail -1 >> $OUT/summary.txt
python3 $ROOT/tools/summarize_prof.py $OUT >> $OUT/summary.txt 2>&1
find $OUT -name "*.csv" -size +2M -delete
tail -3 $OUT/trace.log | cut -c1-300
