export PYTHONPATH=/root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "Crab or Spider or catalogue or every_frame" > gpurun_out/gpu_tests_crab.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_tests_crab.log
[ $rc -ne 0 ] && exit $rc
for l in 20 16 10 8 32; do echo -n "LANES=$l "; BLCD_LANES=$l timeout -k 10 200 python tools/quick_bench.py Crab 20000 200 1 || exit 1; done
for l in 32 20 10; do echo -n "LANES=$l "; BLCD_LANES=$l timeout -k 10 300 python tools/quick_bench.py Crab 40000 200 1 || exit 1; done
for l in 16 8 4; do echo -n "LANES=$l "; BLCD_LANES=$l timeout -k 10 200 python tools/quick_bench.py Crab 4096 200 1 || exit 1; done
