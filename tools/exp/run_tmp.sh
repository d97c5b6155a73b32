for early in 0 1 0 1; do
echo "ZG_HK_LONG_EARLY=$early"; ZG_HK_LONG_EARLY=$early python3 tools/exp/open_tableless.py 20 0 | grep open | tail -3
done
ZG_HK_LONG_EARLY=1 timeout 900 python3 -m pytest tests/test_gpu_api_mirror.py -x -q -m gpu -k "open" 2>&1 | tail -1
