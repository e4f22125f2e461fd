export PYTHONPATH=/root/repo
mkdir -p gpurun_out
BLCD_LIB=libboxlcd_hip_wt.so timeout -k 10 200 python tools/chunk_waves.py Dropbox 100000 50 2 || exit 1
BLCD_LIB=libboxlcd_hip_wt.so timeout -k 10 200 python tools/chunk_waves.py Dropbox 100000 10 6 || exit 1
BLCD_LIB=libboxlcd_hip_wt.so timeout -k 10 200 python tools/chunk_waves.py Object2 200000 10 4 || exit 1
tools/timeline.sh object2_200k Object2 200000 1 || exit 1
for k in "0:0" "50:25" "60:20" "50:10" "30:15"; do BLCD_CHUNK0=$k timeout -k 10 100 python tools/quick_bench.py Dropbox 100000 200 10 || exit 1; done
