mkdir -p gpurun_out
for uses in 0 1; do
timeout 600 ./tools/bench_prove_path synth 20 3 $uses > gpurun_out/r5w_prove_path_uses$uses.json
python3 - $uses <<'PY'
import json,sys
d=json.load(open('gpurun_out/r5w_prove_path_uses%s.json'%sys.argv[1]))['prove_path']
print('uses',sys.argv[1], d['total_ms'], d['total_ms_without_proving_key'], 'key', round(d['steps'][0]['ms'],2), 'commits',round(sum(s['ms'] for s in d['steps'][2:5]),2), 'open', round(d['steps'][-1]['ms'],2))
PY
done
