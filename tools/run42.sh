export PYTHONPATH=/root/repo
for k in "100,100" "100,50,50" "67,67,66" "120,80" "80,120" "100,34,33,33" "50,50,100" "60,140" "140,60"; do echo -n "LIST=$k :: "; BLCD_CHUNK_LIST=$k timeout -k 10 100 python tools/quick_bench.py Bounce 100000 200 20 || exit 1; done
