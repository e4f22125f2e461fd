export PYTHONPATH=/root/repo
for k in "BLCD_TW_SLOTS=1024" "BLCD_TW_SLOTS=2048" "BLCD_TW_SLOTS=1536" "BLCD_TW_SLOTS=2048 BLCD_TWO_WIDTHS=8" "BLCD_TW_SLOTS=2048 BLCD_COHORTS=3" "BLCD_TW_SLOTS=4096"; do echo -n "$k :: "; env $k timeout -k 10 100 python tools/quick_bench.py Object2 200000 200 2 || exit 1; done
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cohorts_of" 2>&1 | tail -2
