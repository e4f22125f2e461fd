export PYTHONPATH=/root/repo
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_tests.log
[ $rc -ne 0 ] && { grep -n "Error\|assert \|FAILED" gpurun_out/gpu_tests.log | head -20; exit $rc; }
timeout -k 10 100 python tools/emit_cost.py Bounce 100000 || exit 1
timeout -k 10 100 python tools/emit_cost.py Dropbox 100000 || exit 1
for rep in 1 2; do timeout -k 10 100 python tools/quick_bench.py Bounce 100000 200 20 || exit 1; done
timeout -k 10 100 python tools/quick_bench.py Object2 200000 200 3 || exit 1
