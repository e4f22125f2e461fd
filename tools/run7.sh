export PYTHONPATH=/root/repo
echo "== old layout (64-lane LDS blocks, one wave per SIMD): libboxlcd_hip_w2.so"
for l in 64 48 32; do echo -n "LANES=$l "; BLCD_LIB=libboxlcd_hip_w2.so BLCD_LANES=$l timeout -k 10 200 python tools/quick_bench.py Crab 20000 200 1 || exit 1; done
for l in 64; do echo -n "LANES=$l "; BLCD_LIB=libboxlcd_hip_w2.so BLCD_LANES=$l timeout -k 10 300 python tools/quick_bench.py Crab 40000 200 1 || exit 1; done
echo "== new layout"
for l in 32; do echo -n "LANES=$l "; BLCD_LANES=$l timeout -k 10 200 python tools/quick_bench.py Crab 4096 200 1 || exit 1; done
for l in 32; do echo -n "LANES=$l "; BLCD_LANES=$l timeout -k 10 300 python tools/quick_bench.py Crab 65536 200 1 || exit 1; done
