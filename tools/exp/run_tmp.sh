mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests/test_gpu_msm.py tests/test_gpu_api_mirror.py -x -q -m gpu 2>&1 | tail -2
for wf in 0 3; do
echo "ZG_MSM_ROWCOL_WAVE_FROM=$wf"
ZG_MSM_ROWCOL_WAVE_FROM=$wf python3 tools/exp/open_tableless.py 20 1 | tail -6
ZG_MSM_ROWCOL_WAVE_FROM=$wf python3 tools/exp/open_tableless.py 20 0 | tail -6
done
