for L in 1 2 4 8 16; do
echo "precompute_levels=$L: $(python bench.py --logn 20 --steps 10 --warmup 2 --precompute $L --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3), d['extra']['kernel_ms_per_msm'], round(d['extra']['setup_seconds'],2))")"
done
