export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out/prof_small
cat > /tmp/small.py <<'PY'
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ["ROOT"])
from zolt_amd import api, lib
lib.init(0)
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
n = 1 << 17
g = api.generator()
ks = np.zeros((n, 4), dtype=np.uint64); ks[:, 0] = np.arange(1, n + 1, dtype=np.uint64)
bases_xy, _ = lib.g1_scalar_mul_batch(np.repeat(g[None, :], n, axis=0), np.zeros(n, dtype=np.uint8), lib.field_op(lib.FR, lib.OP_TO_MONT, ks))
rng = np.random.default_rng(1)
sc = torch.from_numpy(lib.field_op(lib.FR, lib.OP_TO_MONT, rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)).view(np.int64)).to(dev)
d_b = torch.from_numpy(bases_xy.view(np.int64)).to(dev)
b = lib.Bases.upload_dev(d_b.data_ptr(), 0, n, stream=st.cuda_stream)
out = torch.zeros(9, dtype=torch.int64, device=dev)
for _ in range(10):
    b.msm_dev_async(sc.data_ptr(), n, out.data_ptr(), out[8:].data_ptr(), stream=st.cuda_stream)
torch.cuda.synchronize()
PY
cd /tmp
ROOT=$ROOT rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/prof_small/trace -- python3 /tmp/small.py > $ROOT/gpurun_out/prof_small/trace.log 2>&1
cd $ROOT
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_small/trace/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'msm_' in r['Kernel_Name'] and 'precompute' not in r['Kernel_Name']]
rows=rows[-12:]
t0=int(rows[0]['Start_Timestamp'])
for r in rows:
    print(r['Kernel_Name'].split('(')[0].replace('zg::','')[:34].ljust(36), 'start %8.1f us'%((int(r['Start_Timestamp'])-t0)/1e3), 'dur %7.1f us'%((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
PY
rm -rf gpurun_out/prof_small/trace
