export PYTHONPATH=/root/repo
for cfg in "BLCD_TWO_WIDTHS=0" "BLCD_TWO_WIDTHS=16" "BLCD_TWO_WIDTHS=32" "BLCD_TWO_WIDTHS=8"; do
  echo -n "$cfg :: "; env $cfg timeout -k 10 120 python tools/quick_bench.py Bounce 100000 200 5 || exit 1
done
for cfg in "BLCD_TWO_WIDTHS=0" "BLCD_TWO_WIDTHS=16" "BLCD_TWO_WIDTHS=16 BLCD_CHUNK=100" "BLCD_TWO_WIDTHS=16 BLCD_CHUNK=50"; do
  echo -n "$cfg :: "; env $cfg timeout -k 10 120 python tools/quick_bench.py Dropbox 100000 200 3 || exit 1
done
for e in "Object2 200000" "Urchin 50000" "LuxoBall 50000" "Bounce2 100000" "Object3 100000"; do
  echo -n "default :: "; timeout -k 10 200 python tools/quick_bench.py $e 200 2 || exit 1
done
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
