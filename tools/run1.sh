export PYTHONPATH=/root/repo
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; rc=$?
tail -5 gpurun_out/gpu_tests.log
[ $rc -ne 0 ] && exit $rc
tools/fetch_calib.sh && \
tools/timeline.sh dropbox100k Dropbox 100000 2 && \
timeout -k 10 200 python tools/quick_bench.py Crab 20000 200 1 && \
timeout -k 10 100 python tools/quick_bench.py Crab 4096 200 1 && \
timeout -k 10 100 python tools/quick_bench.py CrabCube 20000 200 1
