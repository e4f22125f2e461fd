export PYTHONPATH=/root/repo
for rep in 1 2; do
  timeout -k 10 100 python tools/quick_bench.py Object2 200000 200 2 || exit 1
  timeout -k 10 100 python tools/quick_bench.py Bounce2 100000 200 3 || exit 1
  timeout -k 10 100 python tools/quick_bench.py Object3 100000 200 2 || exit 1
done
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "Object or Bounce2 or every_frame or full_size or two_wave or cohorts or catalogue" > gpurun_out/gpu_tests_o2.log 2>&1; tail -2 gpurun_out/gpu_tests_o2.log
