export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out/prof_sc
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_sc/trace -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $ROOT/gpurun_out/prof_sc/trace.log 2>&1
cd $ROOT
python3 tools/summarize_prof.py gpurun_out/prof_sc | cut -c1-160 | head -40
find gpurun_out/prof_sc -name "*.csv" -size +2M -delete
