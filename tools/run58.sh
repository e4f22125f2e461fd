export PYTHONPATH=/root/repo
for c in 10 20 40; do echo -n "CHUNK=$c :: "; BLCD_CHUNK=$c timeout -k 10 200 python tools/quick_bench.py Object3 100000 200 2 || exit 1; done
for c in 50 100 200; do echo -n "CHUNK=$c :: "; BLCD_CHUNK=$c timeout -k 10 200 python tools/quick_bench.py Bounce2 100000 200 3 || exit 1; done
for c in 10 15 20; do echo -n "CHUNK=$c :: "; BLCD_CHUNK=$c timeout -k 10 200 python tools/quick_bench.py Object2 200000 200 2 || exit 1; done
for e in "UrchinBall 50000" "UrchinBalls 20000" "LuxoCube 50000" "Boxes 100000"; do echo -n "default :: "; timeout -k 10 300 python tools/quick_bench.py $e 200 1 || echo "(failed $e)"; done
