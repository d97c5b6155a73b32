for S in 2 4 8 16; do
  echo "S=$S: $(ZG_MSM_SLICES=$S python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3), d['extra']['kernel_ms_per_msm'])")"
done
