#!/bin/bash
# per-kernel durations (rocprofv3 --kernel-trace --stats) of single MSMs at the sizes given, serial; summaries into gpurun_out/prof_sizes/
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_sizes
export TMPDIR=/tmp
mkdir -p $OUT
cd /tmp
for logn in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t$logn -- python3 $ROOT/tools/bench_tail.py --logn $logn --streams 1 --reps 20 </dev/null > $OUT/t$logn.log 2>&1
  f=$(find $OUT/t$logn -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/kernel_stats_2^$logn.csv
  find $OUT/t$logn -name "*.csv" -size +2M -delete
done
python3 - "$OUT" <<'PY'
import csv, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/kernel_stats_*.csv")):
    print("==", f.split("/")[-1])
    for r in list(csv.DictReader(open(f)))[:24]:
        print(f"{r['Name'][:60]:60s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} min_us={float(r['MinNs'])/1e3:9.1f}")
PY
