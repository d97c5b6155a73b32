for NT in 131072 122880 114688 98304; do
 for ST in 2 3; do
  echo "NT=$NT streams=$ST: $(ZG_MSM_LANES=$ST ZG_MSM_CHUNK_THREADS=$NT python bench.py --steps 30 --warmup 4 --streams $ST --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3))")"
 done
done
