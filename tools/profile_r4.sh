#!/bin/bash
# rocprofv3 recipe for round 4's committed profiles (MSM passes as in round 3; the sumcheck family at one size: tools/profile_r4_sc.sh)
# rocprofv3 recipe for the round's committed profiles (run on the GPU box from the repo root):
#   bash tools/profile_r4.sh <tag>
# pass 1: kernel trace + stats of the SERIAL bench (one stream: every kernel's duration is its own); passes 2..4: PMC counters
# alone (never combined with tracing domains); pass 5: the three-stream bench's kernel stats; then the HBM access-pattern
# calibration (tools/microbench gather|stream under --pmc FETCH_SIZE) and the sumcheck kernels.
TAG=${1:-r4}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
export TMPDIR=/tmp
mkdir -p $OUT
cd /tmp
B1="python3 $ROOT/bench.py --steps 6 --warmup 2 --msms-per-step 4 --streams 1 --no-cpu-baseline --no-extra"
B3="python3 $ROOT/bench.py --steps 10 --warmup 2 --msms-per-step 8 --streams 3 --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1 -- $B1 </dev/null > $OUT/trace1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B1 </dev/null > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B1 </dev/null > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/pmc_sq -- $B1 </dev/null > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace3 -- $B3 </dev/null > $OUT/trace3.log 2>&1
for mode in gather stream; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/cal_$mode -- $ROOT/tools/microbench $mode </dev/null > $OUT/cal_$mode.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_sc -- $ROOT/tools/bench_sumcheck 20 5 </dev/null > $OUT/trace_sc.log 2>&1
cd $OUT
{
  echo "# serial bench command: $B1"; grep -h '^{' $OUT/trace1.log | tail -1
  echo "# three-stream bench command: $B3"; grep -h '^{' $OUT/trace3.log | tail -1
  python3 $ROOT/tools/summarize_prof.py $OUT
  echo "== HBM access-pattern calibration (tools/microbench gather|stream under --pmc FETCH_SIZE) =="
  grep -h "rows of 64 B" $OUT/cal_gather.log $OUT/cal_stream.log
} > $OUT/summary.txt 2>&1
# the metric's second size: kernel stats and HBM counters of the serial 2^22 bench (point slices: four accumulate launches per MSM)
O22=$ROOT/gpurun_out/prof_${TAG}_2e22
mkdir -p $O22
cd /tmp
B22="python3 $ROOT/bench.py --logn 22 --steps 3 --warmup 1 --msms-per-step 2 --streams 1 --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --stats --output-format csv -d $O22/trace1 -- $B22 </dev/null > $O22/trace1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O22/pmc_fetch -- $B22 </dev/null > $O22/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O22/pmc_write -- $B22 </dev/null > $O22/pmc_write.log 2>&1
{
  echo "# serial 2^22 bench command: $B22"; grep -h '^{' $O22/trace1.log | tail -1
  python3 $ROOT/tools/summarize_prof.py $O22
} > $O22/summary.txt 2>&1
find $O22 -name "*.csv" -size +2M -delete
cd $OUT
find $OUT -name "*.csv" -size +2M -delete
tail -c 600 $OUT/trace1.log
# the sumcheck-family kernels at 2^20 entries: durations + FETCH_SIZE / WRITE_SIZE / SQ counters in passes of their own
bash $ROOT/tools/profile_r4_sc.sh $TAG 20
