export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out/prof_sc
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_sc/trace -- $ROOT/tools/bench_sumcheck 20 10 > $ROOT/gpurun_out/prof_sc/trace.log 2>&1
cd $ROOT
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_sc/trace/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(r['Name'].split('(')[0][:44].ljust(46), r['Calls'].rjust(5), ("%.1f"%(float(r['AverageNs'])/1e3)).rjust(9),'us avg', ("%.1f"%(float(r['MaxNs'])/1e3)).rjust(9),'us max')
PY
find gpurun_out/prof_sc -name "*.csv" -size +2M -delete
