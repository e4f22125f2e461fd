export PYTHONPATH=/root/repo
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_tests.log
[ $rc -ne 0 ] && { grep -n "Error\|assert \|FAILED" gpurun_out/gpu_tests.log | head -20; exit $rc; }
timeout -k 10 100 python tools/quick_bench.py Dropbox 100000 200 10 || exit 1
BLCD_COHORTS=1 timeout -k 10 100 python tools/quick_bench.py Dropbox 100000 200 10 || exit 1
timeout -k 10 100 python tools/quick_bench.py Bounce 100000 200 10 || exit 1
BLCD_COHORTS=1 timeout -k 10 100 python tools/quick_bench.py Bounce 100000 200 10 || exit 1
timeout -k 10 100 python tools/quick_bench.py Object2 200000 200 2 || exit 1
tools/timeline.sh dropbox100k_d Dropbox 100000 2 > /dev/null; grep "step_kernel\|span" gpurun_out/dropbox100k_d_timeline.txt | tail -12
