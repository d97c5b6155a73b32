for ST in 1 2 3; do
  echo "streams=$ST: $(ZG_MSM_LANES=$ST python bench.py --steps 20 --warmup 3 --streams $ST --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3), d['extra']['kernel_ms_per_msm'])")"
done
echo "2^22: $(python bench.py --logn 22 --steps 6 --warmup 2 --streams 2 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3), d['extra']['kernel_ms_per_msm'], d['extra']['setup_seconds'])")"
