export PYTHONPATH=/root/repo
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_tests.log
[ $rc -ne 0 ] && exit $rc
tools/timeline.sh dropbox100k_b Dropbox 100000 2 || exit 1
for c in 10 15 20 25 30 40 50; do BLCD_CHUNK=$c timeout -k 10 100 python tools/quick_bench.py Dropbox 100000 200 10 || exit 1; done
for c in 50 100; do BLCD_CHUNK=$c timeout -k 10 100 python tools/quick_bench.py Bounce 100000 200 10 || exit 1; done
for c in 5 10 20; do BLCD_CHUNK=$c timeout -k 10 100 python tools/quick_bench.py Object2 200000 200 2 || exit 1; done
