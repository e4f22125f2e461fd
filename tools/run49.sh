export PYTHONPATH=/root/repo
for cfg in "BLCD_UNSORTED_SPREAD=0" "BLCD_UNSORTED_SPREAD=1" "BLCD_UNSORTED_SPREAD=0" "BLCD_UNSORTED_SPREAD=1" "BLCD_COHORTS=3" "BLCD_COHORTS=4" "BLCD_COHORTS=1"; do
  echo -n "$cfg :: "; env $cfg timeout -k 10 120 python tools/quick_bench.py Bounce 100000 200 5 || exit 1
done
for cfg in "BLCD_UNSORTED_SPREAD=0" "BLCD_UNSORTED_SPREAD=1" "BLCD_COHORTS=3" "BLCD_COHORTS=4"; do
  echo -n "$cfg :: "; env $cfg timeout -k 10 120 python tools/quick_bench.py Dropbox 100000 200 5 || exit 1
done
for cfg in "BLCD_UNSORTED_SPREAD=0" "BLCD_UNSORTED_SPREAD=1"; do
  echo -n "$cfg :: "; env $cfg timeout -k 10 120 python tools/quick_bench.py Bounce2 100000 200 3 || exit 1
done
