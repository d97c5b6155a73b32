mkdir -p gpurun_out
for lv in 1 2 3 4 5 8 15; do
timeout 600 ./tools/bench_prove_path synth 20 3 0 $lv > gpurun_out/pp_$lv.json
python3 - $lv <<'PY'
import json,sys
d=json.load(open('gpurun_out/pp_%s.json'%sys.argv[1]))
if 'error' in d: print(sys.argv[1], d); sys.exit()
d=d['prove_path']
st={s['call'][:12]:s['ms'] for s in d['steps']}
print('levels',sys.argv[1], 'total',d['total_ms'], 'key',round(d['steps'][0]['ms'],2), 'commits',round(sum(s['ms'] for s in d['steps'][2:5]),2), 'open',round(d['steps'][-1]['ms'],2))
PY
done
