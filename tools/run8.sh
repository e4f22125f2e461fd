export PYTHONPATH=/root/repo
tools/timeline.sh bounce100k Bounce 100000 2 || exit 1
BLCD_LIB=libboxlcd_hip_wt.so timeout -k 10 200 python tools/chunk_waves.py Bounce 100000 100 2 || exit 1
BLCD_LIB=libboxlcd_hip_wt.so timeout -k 10 200 python tools/chunk_waves.py Bounce 100000 25 8 || exit 1
