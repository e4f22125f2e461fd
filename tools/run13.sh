export PYTHONPATH=/root/repo
for v in k1 k2 k3; do
  echo "== variant $v"
  BLCD_LIB=libboxlcd_hip_$v.so timeout -k 10 100 python tools/quick_bench.py Dropbox 100000 200 10 || exit 1
  BLCD_LIB=libboxlcd_hip_$v.so BLCD_CHUNK=25 timeout -k 10 100 python tools/quick_bench.py Dropbox 100000 200 10 || exit 1
done
BLCD_LIB=libboxlcd_hip_k2.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -x -q -m gpu -k "Dropbox or at_rest or every_frame or baseline_configs or catalogue or cohorts" > gpurun_out/gpu_tests_k2.log 2>&1; tail -3 gpurun_out/gpu_tests_k2.log
BLCD_LIB=libboxlcd_hip_k2.so tools/timeline.sh dropbox100k_k2 Dropbox 100000 2
