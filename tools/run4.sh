export PYTHONPATH=/root/repo
for k in "0:0" "50:10" "50:5" "50:17" "50:25" "60:10" "60:15" "60:20" "40:10" "100:10" "50:10:0"; do echo -n "CHUNK0=$k  "; BLCD_CHUNK0=$k timeout -k 10 100 python tools/quick_bench.py Dropbox 100000 200 10 || exit 1; done
for k in "50:10" "60:10"; do echo -n "CHUNK0=$k COHORTS=3 "; BLCD_COHORTS=3 BLCD_CHUNK0=$k timeout -k 10 100 python tools/quick_bench.py Dropbox 100000 200 10 || exit 1; done
for k in "0:0" "30:5" "30:10:1" "60:5"; do echo -n "Object2 CHUNK0=$k  "; BLCD_CHUNK0=$k timeout -k 10 100 python tools/quick_bench.py Object2 200000 200 2 || exit 1; done
BLCD_CHUNK0=50:10 tools/timeline.sh dropbox100k_c Dropbox 100000 2
