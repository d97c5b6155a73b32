python -m pytest tests/test_gpu_msm.py -m gpu -x -q 2>&1 | tail -5
for SCHED in 1 0; do
  echo "chunk_sched=$SCHED: $(ZG_MSM_CHUNK_SCHED=$SCHED python bench.py --steps 10 --warmup 2 --streams 1 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3), d['extra']['kernel_ms_per_msm'])")"
done
echo "streams2: $(python bench.py --steps 20 --warmup 3 --streams 2 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3), d['extra']['kernel_ms_per_msm'])")"
