#!/bin/bash
set -u
mkdir -p gpurun_out/c19
OUT=$PWD/gpurun_out/c19
python -m pytest tests/test_gpu_msm.py -x -q -k "every_window_size" 2>&1 | tail -4 > $OUT/pytest.txt
for c in 17 18 19; do
  python bench.py --logn 20 --steps 10 --warmup 2 --msms-per-step 16 --window-bits $c --no-cpu-baseline --no-extra > $OUT/l20_c$c.json 2> $OUT/l20_c$c.err
  python bench.py --logn 20 --steps 4 --warmup 1 --msms-per-step 8 --window-bits $c --no-cpu-baseline --no-extra --streams 1 > $OUT/l20_c${c}_serial.json 2>> $OUT/l20_c$c.err
done
