export PYTHONPATH=/root/repo
for cfg in "A=1" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=8 BLCD_COHORTS=3" "GPU_MAX_HW_QUEUES=8 BLCD_COHORTS=4" "GPU_MAX_HW_QUEUES=8 BLCD_COHORTS=4 BLCD_CHUNK=20"; do
  echo -n "$cfg :: "; env $cfg timeout -k 10 200 python tools/quick_bench.py Object2 200000 200 2 || exit 1
done
for cfg in "GPU_MAX_HW_QUEUES=8 BLCD_COHORTS=3" "GPU_MAX_HW_QUEUES=8 BLCD_COHORTS=4"; do
  echo -n "$cfg :: "; env $cfg timeout -k 10 200 python tools/quick_bench.py Bounce 100000 200 5 || exit 1
  echo -n "$cfg :: "; env $cfg timeout -k 10 200 python tools/quick_bench.py Dropbox 100000 200 3 || exit 1
done
