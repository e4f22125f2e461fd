export PYTHONPATH=/root/repo
for k in "BLCD_CHUNK=67" "BLCD_CHUNK=100" "BLCD_CHUNK=200" "BLCD_CHUNK=100 BLCD_COHORTS=1" "BLCD_CHUNK=200 BLCD_COHORTS=1" "BLCD_CHUNK=50"; do echo -n "$k :: "; env $k timeout -k 10 100 python tools/quick_bench.py Bounce 100000 200 20 || exit 1; done
for k in "BLCD_CHUNK=10" "BLCD_CHUNK=20" "BLCD_CHUNK=50" "BLCD_CHUNK=100"; do echo -n "$k :: "; env $k timeout -k 10 100 python tools/quick_bench.py Bounce2 100000 200 3 || exit 1; done
