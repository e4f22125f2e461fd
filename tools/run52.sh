export PYTHONPATH=/root/repo
export BLCD_LIB=libboxlcd_hip_pt.so CW_RAW=1
timeout -k 10 200 python tools/chunk_waves.py Bounce 100000 100 2 || exit 1
timeout -k 10 200 python tools/chunk_waves.py Dropbox 100000 200 1 || exit 1
