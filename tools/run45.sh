export PYTHONPATH=/root/repo
export BLCD_LIB=libboxlcd_hip_wt.so
timeout -k 10 200 python tools/chunk_waves.py Dropbox 100000 200 1 || exit 1
timeout -k 10 200 python tools/chunk_waves.py Dropbox 100000 50 4 || exit 1
timeout -k 10 200 python tools/chunk_waves.py Object2 200000 10 3 || exit 1
