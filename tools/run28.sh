export PYTHONPATH=/root/repo
for rep in 1 2; do for lib in libboxlcd_hip.so libboxlcd_hip_cr.so; do echo -n "$lib "; BLCD_LIB=$lib timeout -k 10 100 python tools/quick_bench.py Bounce 100000 200 20 || exit 1; done; done
BLCD_LIB=libboxlcd_hip_cr.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -x -q -m gpu -k "Bounce and not Bounce2" 2>&1 | tail -2
