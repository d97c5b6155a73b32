#!/bin/bash
# The round's committed rocprofv3 evidence, from ONE command on the GPU box (repo root):
#     bash tools/collect_profiles.sh <tag>          e.g. r5
# writes gpurun_out/prof_<tag>/ and COPIES the judged files into profiles/:
#     profiles/<tag>_rocprofv3_summary.txt         kernel stats + PMC averages, 2^20 serial bench (+ three-stream kernel stats)
#     profiles/<tag>_rocprofv3_summary_2^22.txt    the same for the metric's second size
#     profiles/<tag>_pmc.json                      the numbers bench.py reads for roofline.traffic / valu_issue (tools/pmc_json.py):
#                                                  FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU, stall split and avg duration of msm_accumulate
#     profiles/<tag>_kernel_stats_trace1.csv, _trace3.csv
# Counters are collected in passes of their own (--pmc only, never with a tracing domain); kernel durations come from
# --kernel-trace --stats of the same bench command. The program itself stands behind `--` (python3 bench.py ...), no wrapper.
TAG=${1:-r5}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
export TMPDIR=/tmp
mkdir -p $OUT $ROOT/profiles
cd /tmp
B1="python3 $ROOT/bench.py --steps 6 --warmup 2 --msms-per-step 4 --streams 1 --no-cpu-baseline --no-extra"
B3="python3 $ROOT/bench.py --steps 10 --warmup 2 --msms-per-step 8 --streams 3 --no-cpu-baseline --no-extra"
B22="python3 $ROOT/bench.py --logn 22 --steps 3 --warmup 1 --msms-per-step 2 --streams 1 --no-cpu-baseline --no-extra"
pass() {  # pass <dir> <bench command> <rocprofv3 options...>
  local d=$1 cmd=$2; shift 2
  rocprofv3 "$@" --output-format csv -d $d -- $cmd </dev/null > $d.log 2>&1
}
pass $OUT/trace1 "$B1" --kernel-trace --stats
pass $OUT/pmc_fetch "$B1" --pmc FETCH_SIZE
pass $OUT/pmc_write "$B1" --pmc WRITE_SIZE
pass $OUT/pmc_sq "$B1" --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES
# stall side (review item 6): wave-parked / issue-stalled / active quad-cycles (disjoint, sum ~ SQ_WAVE_CYCLES) and vector-memory instructions
pass $OUT/pmc_stall "$B1" --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS
pass $OUT/pmc_tcp "$B1" --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
pass $OUT/pmc_tcc "$B1" --pmc TCC_HIT_sum TCC_MISS_sum
pass $OUT/trace3 "$B3" --kernel-trace --stats
for mode in gather stream; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/cal_$mode -- $ROOT/tools/microbench $mode </dev/null > $OUT/cal_$mode.log 2>&1
done
{
  echo "# serial bench command: $B1"; grep -h '^{' $OUT/trace1.log | tail -1
  echo "# three-stream bench command: $B3"; grep -h '^{' $OUT/trace3.log | tail -1
  python3 $ROOT/tools/summarize_prof.py $OUT
  echo "== HBM access-pattern calibration (tools/microbench gather|stream under --pmc FETCH_SIZE) =="
  grep -h "rows of 64 B" $OUT/cal_gather.log $OUT/cal_stream.log
} > $OUT/summary.txt 2>&1
O22=$ROOT/gpurun_out/prof_${TAG}_2e22
mkdir -p $O22
pass $O22/trace1 "$B22" --kernel-trace --stats
pass $O22/pmc_fetch "$B22" --pmc FETCH_SIZE
pass $O22/pmc_write "$B22" --pmc WRITE_SIZE
pass $O22/pmc_sq "$B22" --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES
{
  echo "# serial 2^22 bench command: $B22"; grep -h '^{' $O22/trace1.log | tail -1
  python3 $ROOT/tools/summarize_prof.py $O22
} > $O22/summary.txt 2>&1
# the sumcheck-family kernels at ONE size (2^20 entries): durations, then counters in passes of their own
OSC=$ROOT/gpurun_out/prof_sc_$TAG
mkdir -p $OSC
SC="python3 $ROOT/tools/prof_sc_kernels.py 20 8"
pass $OSC/trace "$SC" --kernel-trace --stats
pass $OSC/pmc_fetch "$SC" --pmc FETCH_SIZE
pass $OSC/pmc_write "$SC" --pmc WRITE_SIZE
pass $OSC/pmc_sq "$SC" --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES
{
  echo "# command: $SC   (every kernel below ran on 2^20-entry tables only; FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them:"
  echo "#  FETCH_SIZE counts a wide coalesced streaming read at half its bytes on gfx950 — MI355X_MICROARCH.md — double it before comparing)"
  python3 $ROOT/tools/summarize_prof.py $OSC
} > $OSC/summary.txt 2>&1
cp $OSC/summary.txt "$ROOT/profiles/${TAG}_sumcheck_kernels_2^20_durations_and_pmc.txt"
find $OSC -name "*.csv" -size +2M -delete
python3 $ROOT/tools/pmc_json.py $TAG $OUT $O22 > $OUT/pmc.json
cp $OUT/summary.txt $ROOT/profiles/${TAG}_rocprofv3_summary.txt
cp $O22/summary.txt "$ROOT/profiles/${TAG}_rocprofv3_summary_2^22.txt"
python3 $ROOT/tools/kernel_table.py $TAG > $ROOT/profiles/${TAG}_kernel_table.md 2>/dev/null  # (reads the two summaries copied above)
cp $OUT/pmc.json $ROOT/profiles/${TAG}_pmc.json
for t in trace1 trace3; do
  f=$(find $OUT/$t -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $ROOT/profiles/${TAG}_kernel_stats_$t.csv
done
find $OUT $O22 -name "*.csv" -size +2M -delete
# the judged copies travel back with gpurun_out (profiles/ itself is not merged back from the box)
mkdir -p $ROOT/gpurun_out/profiles_$TAG && cp $ROOT/profiles/${TAG}_* $ROOT/gpurun_out/profiles_$TAG/
tail -c 400 $OUT/trace1.log; cat $OUT/pmc.json | head -c 1500
