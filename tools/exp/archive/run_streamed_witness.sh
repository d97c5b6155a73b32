# streamed witness build at 2^20 cycles: tests, the slice sweep of bench_sumcheck, the prove-path stage-1 line
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_ingest.py tests/test_gpu_cpp_host.py -x -q -m gpu 2>&1 | tail -3
timeout 600 ./tools/bench_sumcheck 20 3 > gpurun_out/r5i_bench_sumcheck.json 2> gpurun_out/r5i_bench_sumcheck.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r5i_bench_sumcheck.json').read())
for k in d:
    if k.startswith('outer'): print(k, d[k])
PY
for i in 1 2; do
timeout 600 ./tools/bench_prove_path synth 20 3 > gpurun_out/r5i_prove_path.json
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r5i_prove_path.json'))['prove_path']
print(d['total_ms'], d['total_ms_without_proving_key'])
for s in d['steps']:
    if s['call'].startswith('stage 1: trace'): print('stage-1 witness', round(s['ms'],3))
PY
done
