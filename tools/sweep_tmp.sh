export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out/prof_22
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_22/trace -- python3 $ROOT/bench.py --logn 22 --steps 6 --warmup 2 --streams 1 --no-cpu-baseline --no-extra > $ROOT/gpurun_out/prof_22/trace.log 2>&1
cd $ROOT
python3 tools/summarize_prof.py gpurun_out/prof_22 | cut -c1-150 | head -22
find gpurun_out/prof_22 -name "*.csv" -size +2M -delete
