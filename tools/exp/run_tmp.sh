mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests/test_gpu_poly.py tests/test_gpu_handoff.py tests/test_gpu_cpp_host.py -x -q -m gpu 2>&1 | tail -3
ZG_SC_TAIL_MAX=1 timeout 900 python3 -m pytest tests/test_gpu_poly.py -x -q -m gpu -k "device_resident or run_sumcheck" 2>&1 | tail -1
ZG_SC_TAIL_MAX=64 ZG_SC_MAX_BLOCKS=3 timeout 900 python3 -m pytest tests/test_gpu_poly.py -x -q -m gpu -k "device_resident or run_sumcheck" 2>&1 | tail -1
timeout 600 ./tools/bench_sumcheck 20 20 > gpurun_out/r5r_bench_sumcheck.json 2>/dev/null
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r5r_bench_sumcheck.json').read())
for k in ('device_resident_rounds_per_s','device_resident_ms_runSumcheck','rounds_per_s','us_per_round','ms_runSumcheck'): print(k, d.get(k))
PY
