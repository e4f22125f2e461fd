export PYTHONPATH=/root/repo
for cfg in "BLCD_TWO_WIDTHS=4" "BLCD_TWO_WIDTHS=6" "BLCD_TWO_WIDTHS=8" "BLCD_TWO_WIDTHS=12" "BLCD_TWO_WIDTHS=8 BLCD_TW_SLOTS=2048" "BLCD_TWO_WIDTHS=16 BLCD_TW_SLOTS=2048" "BLCD_TWO_WIDTHS=8 BLCD_TW_SLOTS=512" "BLCD_TWO_WIDTHS=8 BLCD_CHUNK=50" "BLCD_TWO_WIDTHS=8 BLCD_CHUNK=67"; do
  echo -n "$cfg :: "; env $cfg timeout -k 10 120 python tools/quick_bench.py Bounce 100000 200 5 || exit 1
done
for cfg in "BLCD_TWO_WIDTHS=0" "BLCD_TWO_WIDTHS=16" "BLCD_TWO_WIDTHS=8" "BLCD_TWO_WIDTHS=0" "BLCD_TWO_WIDTHS=16"; do
  echo -n "$cfg :: "; env $cfg timeout -k 10 120 python tools/quick_bench.py Dropbox 100000 200 5 || exit 1
done
for e in "Object2 200000" "Bounce2 100000" "Object3 100000"; do
  for cfg in "BLCD_TWO_WIDTHS=8" "BLCD_TWO_WIDTHS=16 BLCD_TW_SLOTS=2048" "BLCD_TWO_WIDTHS=32"; do
    echo -n "$cfg :: "; env $cfg timeout -k 10 200 python tools/quick_bench.py $e 200 2 || exit 1
  done
done
