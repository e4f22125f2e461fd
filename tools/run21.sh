export PYTHONPATH=/root/repo
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_tests.log
[ $rc -ne 0 ] && { grep -n "Error\|assert \|FAILED" gpurun_out/gpu_tests.log | head -20; exit $rc; }
for c in 30 40 50 67 100; do echo -n "CHUNK=$c "; BLCD_CHUNK=$c timeout -k 10 100 python tools/quick_bench.py Dropbox 100000 200 10 || exit 1; done
bash tools/profile_all.sh > gpurun_out/profile_all.log 2>&1; tail -3 gpurun_out/profile_all.log
