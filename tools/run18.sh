export PYTHONPATH=/root/repo
for rep in 1 2; do timeout -k 10 100 python tools/quick_bench.py Bounce 100000 200 20 || exit 1; done
for c in 5 10 20; do for k in 1 2; do echo -n "CHUNK=$c COHORTS=$k "; BLCD_CHUNK=$c BLCD_COHORTS=$k timeout -k 10 100 python tools/quick_bench.py Object2 200000 200 2 || exit 1; done; done
for w in 0 16 32; do echo -n "TWO_WIDTHS=$w "; BLCD_TWO_WIDTHS=$w timeout -k 10 100 python tools/quick_bench.py Object2 200000 200 2 || exit 1; done
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "Object2 or Bounce or every_frame or full_size or two_wave or cohorts" > gpurun_out/gpu_tests_o2.log 2>&1; tail -2 gpurun_out/gpu_tests_o2.log
