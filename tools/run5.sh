export PYTHONPATH=/root/repo
echo "== default library"
timeout -k 10 100 python tools/quick_bench.py Dropbox 100000 200 10 || exit 1
timeout -k 10 100 python tools/quick_bench.py Urchin 50000 200 2 || exit 1
timeout -k 10 200 python tools/quick_bench.py Urchin 100000 200 2 || exit 1
timeout -k 10 200 python tools/quick_bench.py Urchin 130000 200 2 || exit 1
echo "== 2 waves per SIMD variant (256 VGPRs: 668 / 1033 spills)"
export BLCD_LIB=libboxlcd_hip_w2.so
timeout -k 10 100 python tools/quick_bench.py Dropbox 100000 200 10 || exit 1
timeout -k 10 100 python tools/quick_bench.py Urchin 50000 200 2 || exit 1
BLCD_LANES=64 timeout -k 10 100 python tools/quick_bench.py Urchin 50000 200 2 || exit 1
BLCD_LANES=32 timeout -k 10 100 python tools/quick_bench.py Urchin 50000 200 2 || exit 1
timeout -k 10 200 python tools/quick_bench.py Urchin 100000 200 2 || exit 1
timeout -k 10 200 python tools/quick_bench.py Urchin 130000 200 2 || exit 1
