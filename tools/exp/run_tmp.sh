mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests/test_gpu_poly.py tests/test_gpu_cpp_host.py tests/test_gpu_prover_sites.py -x -q -m gpu 2>&1 | tail -3
for uses in 0 1; do
timeout 600 ./tools/bench_prove_path synth 20 3 $uses > gpurun_out/pp_$uses.json
python3 - $uses <<'PY'
import json,sys
d=json.load(open('gpurun_out/pp_%s.json'%sys.argv[1]))
if 'error' in d: print(d); sys.exit()
d=d['prove_path']
print('uses',sys.argv[1], d['total_ms'], d['total_ms_without_proving_key'])
for s in d['steps']:
    if 'stage 1' in s['call']: print('  ',round(s['ms'],3), s['call'][:80])
PY
done
