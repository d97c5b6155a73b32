mkdir -p gpurun_out
timeout 1200 python3 bench.py > gpurun_out/r5l_bench.json 2> gpurun_out/r5l_bench.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r5l_bench.json').read().strip().splitlines()[-1])
e=d['extra']
print(d['value'], d['ms_per_step'], e['single_process_c_abi']['none']['pipelined']['msm_per_s'], e['msm_2^22_single_gpu']['value'], d['config']['table_build_ms'])
print('prove_path', e['prove_path'].get('total_ms'), e['prove_path'].get('total_ms_without_proving_key'), e['prove_path']['steps'][0]['ms'])
print('single use', e['prove_path_single_use_key']['total_ms'], e['prove_path_single_use_key']['steps_ms'])
PY
