#!/bin/bash
# copies what tools/exp/final_pass_r4.sh <tag> left under gpurun_out/ into profiles/ under the names the docs cite, then regenerates the
# kernel table:  bash tools/collect_profiles.sh r4final
set -u
TAG=${1:-r4final}
G=gpurun_out
P=profiles
cp $G/$TAG/bench.json $P/${TAG}_bench.json
cp $G/$TAG/bench_sumcheck.json $P/${TAG}_bench_sumcheck.json
cp $G/$TAG/bench_sumcheck_v13.json $P/${TAG}_bench_sumcheck_v13.json
cp $G/$TAG/pytest_gpu_tail.txt $P/${TAG}_pytest_gpu_tail.txt
cp $G/$TAG/stage3_round_split.txt $P/${TAG}_stage3_round_split.txt
cp $G/prof_$TAG/summary.txt $P/${TAG}_rocprofv3_summary.txt
cp $G/prof_${TAG}_2e22/summary.txt $P/${TAG}_rocprofv3_summary_2^22.txt
cp $G/prof_sc_$TAG/summary.txt "$P/${TAG}_sumcheck_kernels_2^20_durations_and_pmc.txt"
for t in trace1 trace3 trace_sc; do
  f=$(find $G/prof_$TAG/$t -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $P/${TAG}_kernel_stats_$t.csv
done
f=$(find $G/prof_${TAG}_2e22/trace1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f "$P/${TAG}_kernel_stats_2^22_serial.csv"
f=$(find $G/prof_sc_$TAG/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f "$P/${TAG}_kernel_stats_sumcheck_2^20.csv"
python3 tools/kernel_table.py $TAG > $P/${TAG}_kernel_table.md
ls -la $P | grep ${TAG}_ | wc -l
