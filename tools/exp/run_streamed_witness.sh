# A/B: host threads and slice counts of CycleWitnessMatrix::fromTrace at 2^20 cycles (prove-path composite, stage-1 line)
mkdir -p gpurun_out
for th in 8 16 32; do for sl in 2 4 8; do
  ZOLT_HOST_THREADS=$th ZOLT_WITNESS_SLICES=$sl timeout 600 ./tools/bench_prove_path synth 20 3 > gpurun_out/pp.json
  python3 - "$th" "$sl" <<'PY'
import json,sys
d=json.load(open('gpurun_out/pp.json'))['prove_path']
for s in d['steps']:
    if s['call'].startswith('stage 1: trace') or s['call'].startswith('commit: host builds'): print('threads',sys.argv[1],'slices',sys.argv[2], round(s['ms'],3), s['call'][:40])
PY
done; done 2>&1 | tee gpurun_out/r5h_streamed_witness_ab.txt
timeout 600 ./tools/bench_prove_path synth 20 3 > gpurun_out/pp.json
python3 - default default <<'PY'
import json,sys
d=json.load(open('gpurun_out/pp.json'))['prove_path']
print(d['total_ms'], d['total_ms_without_proving_key'])
for s in d['steps']:
    if s['call'].startswith('stage 1: trace'): print('default', round(s['ms'],3))
PY
timeout 300 python3 -m pytest tests/test_gpu_cpp_host.py -x -q -m gpu -k "witness or composite" 2>&1 | tail -2
