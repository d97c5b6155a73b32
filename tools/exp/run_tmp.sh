mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests/test_gpu_msm.py -x -q -m gpu 2>&1 | tail -3
for v1 in 1 0; do
ZG_MSM_PRECOMPUTE_V1=$v1 timeout 600 ./tools/bench_prove_path synth 20 3 0 > gpurun_out/pp_$v1.json
python3 - $v1 <<'PY'
import json,sys
d=json.load(open('gpurun_out/pp_%s.json'%sys.argv[1]))
if 'error' in d: print(d); sys.exit()
d=d['prove_path']
print('v1',sys.argv[1], d['total_ms'], d['total_ms_without_proving_key'], 'key',round(d['steps'][0]['ms'],2), 'cold key', round(d['steps'][0]['ms_cold'],2))
PY
done
