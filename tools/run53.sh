export PYTHONPATH=/root/repo
for cfg in "A=1" "BLCD_SORT_FIRST=1" "BLCD_SORT_FIRST=1 BLCD_CHUNK=50" "BLCD_SORT_FIRST=1 BLCD_CHUNK=67" "A=1"; do
  echo -n "$cfg :: "; env $cfg timeout -k 10 120 python tools/quick_bench.py Bounce 100000 200 5 || exit 1
done
