export PYTHONPATH=/root/repo
BLCD_LIB=libboxlcd_hip_t1.so timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/gpu_tests_t1.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_tests_t1.log
[ $rc -ne 0 ] && { grep -n "Error\|assert \|FAILED" gpurun_out/gpu_tests_t1.log | head; exit $rc; }
for lib in libboxlcd_hip.so libboxlcd_hip_t1.so; do
  echo "== $lib"
  BLCD_LIB=$lib timeout -k 10 100 python tools/quick_bench.py Object2 200000 200 3 || exit 1
  BLCD_LIB=$lib timeout -k 10 100 python tools/quick_bench.py Bounce2 100000 200 3 || exit 1
  BLCD_LIB=$lib timeout -k 10 100 python tools/quick_bench.py Object3 100000 200 2 || exit 1
  BLCD_LIB=$lib timeout -k 10 100 python tools/quick_bench.py Urchin 50000 200 2 || exit 1
  BLCD_LIB=$lib timeout -k 10 100 python tools/quick_bench.py LuxoBall 50000 200 2 || exit 1
  BLCD_LIB=$lib timeout -k 10 100 python tools/quick_bench.py UrchinBalls 20000 200 1 || exit 1
done
