mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests/test_gpu_msm.py tests/test_gpu_api_mirror.py -x -q -m gpu -k "fixed or setup or hyperkzg" 2>&1 | tail -2
for c in 8 10 11 12 13; do
echo "ZG_FB_WINDOW_BITS=$c"
ZG_FB_WINDOW_BITS=$c timeout 600 ./tools/bench_prove_path synth 20 3 1 > gpurun_out/pp.json
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/pp.json'))['prove_path']
print('  key', round(d['steps'][0]['ms'],3))
PY
done
ZG_FB_WINDOW_BITS=5 timeout 600 python3 -m pytest tests/test_gpu_msm.py -x -q -m gpu -k "fixed or setup" 2>&1 | tail -1
