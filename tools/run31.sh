export PYTHONPATH=/root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cohorts or two_wave or full_size or ragged or every_frame or chunking or rest" > gpurun_out/gpu_tests_rb.log 2>&1; tail -2 gpurun_out/gpu_tests_rb.log
for rep in 1 2 3; do timeout -k 10 100 python tools/quick_bench.py Bounce 100000 200 20 || exit 1; done
timeout -k 10 100 python tools/quick_bench.py Object2 200000 200 2 || exit 1
timeout -k 10 100 python tools/quick_bench.py Dropbox 100000 200 10 || exit 1
tools/timeline.sh bounce100k_b Bounce 100000 2 > /dev/null; grep "rebin\|step_kernel\|span" gpurun_out/bounce100k_b_timeline.txt | head -16
