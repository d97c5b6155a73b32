mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_prover_sites.py tests/test_gpu_cpp_host.py -x -q -m gpu 2>&1 | tail -3
timeout 600 ./tools/bench_prove_path synth 20 3 > gpurun_out/r5j_prove_path.json
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r5j_prove_path.json'))['prove_path']
print(d['total_ms'], d['total_ms_without_proving_key'])
for s in d['steps']: print(round(s['ms'],3), s['call'][:100])
PY
