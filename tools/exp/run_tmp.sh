mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_pp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pp -- $R/tools/bench_prove_path synth 20 3 </dev/null > /tmp/prof_pp.log 2>&1
f=$(find /tmp/prof_pp -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/r5t_prove_path_kernel_stats.csv; head -25 "$f" | cut -c1-150
