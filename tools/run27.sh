export PYTHONPATH=/root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -x -q -m gpu -k "Crab or Spider or catalogue or every_frame or step_obs" > gpurun_out/gpu_tests_crab.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_tests_crab.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python tools/quick_bench.py Crab 20000 200 1 || exit 1
timeout -k 10 200 python tools/quick_bench.py CrabCube 20000 200 1 || exit 1
timeout -k 10 200 python tools/quick_bench.py SpiderCube 20000 200 1 || exit 1
for k in "BLCD_TW_SLOTS=1024" "BLCD_TW_SLOTS=2048" "BLCD_TW_SLOTS=2048 BLCD_TWO_WIDTHS=8" "BLCD_TW_SLOTS=4096"; do echo -n "$k :: "; env $k timeout -k 10 100 python tools/quick_bench.py Object2 200000 200 2 || exit 1; done
