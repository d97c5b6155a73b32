for NT in 131072 196608 262144; do
  echo "NT=$NT: $(ZG_MSM_CHUNK_THREADS=$NT python bench.py --steps 10 --warmup 2 --streams 1 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3), d['extra']['kernel_ms_per_msm'])")"
done
