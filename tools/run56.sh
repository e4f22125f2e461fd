export PYTHONPATH=/root/repo
for lib in libboxlcd_hip.so libboxlcd_hip_pipe1.so libboxlcd_hip_pipe2.so; do
  for e in "Urchin 50000" "LuxoBall 50000"; do
    echo -n "$lib :: "; BLCD_LIB=$lib timeout -k 10 200 python tools/quick_bench.py $e 200 2 || exit 1
  done
done
BLCD_LIB=libboxlcd_hip_pipe1.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4
