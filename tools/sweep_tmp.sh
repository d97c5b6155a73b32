export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out/prof_skew
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY --output-format csv -d $ROOT/gpurun_out/prof_skew/pmc -- python3 $ROOT/tools/bench_skew.py > $ROOT/gpurun_out/prof_skew/pmc.log 2>&1
cd $ROOT
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_skew/pmc/**/*counter_collection.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
acc={}
for r in rows:
    n=r['Kernel_Name'].split('(')[0]
    if 'heavy_a' in n or 'bitsum' in n:
        acc.setdefault((n,r['Dispatch_Id']),{})[r['Counter_Name']]=float(r['Counter_Value'])
for (n,d),c in list(acc.items())[-16:]:
    print(n,d,c)
PY
rm -rf gpurun_out/prof_skew/pmc
