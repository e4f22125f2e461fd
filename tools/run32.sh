export PYTHONPATH=/root/repo
for k in "BLCD_COHORTS=2" "BLCD_COHORTS=3" "BLCD_COHORTS=3 BLCD_TWO_WIDTHS=8" "BLCD_COHORTS=2 BLCD_TWO_WIDTHS=8" "BLCD_COHORTS=3 BLCD_CHUNK=20" "BLCD_COHORTS=3 BLCD_TWO_WIDTHS=8 BLCD_CHUNK=20" "BLCD_COHORTS=3 BLCD_CHUNK=5"; do echo -n "$k :: "; env $k timeout -k 10 100 python tools/quick_bench.py Object2 200000 200 3 || exit 1; done
for k in "BLCD_CHUNK=20" "BLCD_CHUNK=100" "BLCD_CHUNK=100 BLCD_COHORTS=3" "BLCD_CHUNK=200"; do echo -n "$k :: "; env $k timeout -k 10 100 python tools/quick_bench.py Bounce2 100000 200 3 || exit 1; done
for k in "BLCD_CHUNK=20" "BLCD_CHUNK=50" "BLCD_CHUNK=100" "BLCD_CHUNK=20 BLCD_COHORTS=3"; do echo -n "$k :: "; env $k timeout -k 10 100 python tools/quick_bench.py Object3 100000 200 2 || exit 1; done
