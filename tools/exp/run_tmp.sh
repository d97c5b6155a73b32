mkdir -p gpurun_out
timeout 1200 python3 -m pytest tests/test_gpu_api_mirror.py -x -q -m gpu -k "without_a_table or long_levels" 2>&1 | tail -5
for uses in 1 4; do
timeout 600 ./tools/bench_prove_path synth 20 3 $uses > gpurun_out/pp_$uses.json
python3 - $uses <<'PY'
import json,sys
d=json.load(open('gpurun_out/pp_%s.json'%sys.argv[1]))
if 'error' in d: print(d); sys.exit()
d=d['prove_path']
print('uses',sys.argv[1], d['total_ms'], d['total_ms_without_proving_key'], 'key',round(d['steps'][0]['ms'],2), 'commits',round(sum(s['ms'] for s in d['steps'][2:5]),2), 'open',round(d['steps'][-1]['ms'],2))
PY
python3 tools/exp/open_tableless.py 20 $uses
done
