# full validation: GPU suite, smoke, bench (as the driver runs them)
mkdir -p gpurun_out
TAG=${1:-r5k}
timeout 3000 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/${TAG}_pytest_gpu_tail.txt; cat gpurun_out/${TAG}_pytest_gpu_tail.txt
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1200 python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 300 gpurun_out/${TAG}_bench.err
python3 - $TAG <<'PY'
import json,sys
d=json.loads(open('gpurun_out/%s_bench.json'%sys.argv[1]).read().strip().splitlines()[-1])
print(d['value'], d['unit'], d['roofline']['frac'], d['cpu_baseline']['value'])
e=d['extra']
print('prove_path', e['prove_path'].get('total_ms'), e['prove_path'].get('total_ms_without_proving_key'))
print('single use', e.get('prove_path_single_use_key'))
print('open', e['hyperkzg_open'])
PY
