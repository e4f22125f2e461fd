export PYTHONPATH=/root/repo
for rep in 1 2; do
for lib in libboxlcd_hip.so libboxlcd_hip_o2.so; do
  echo "== $lib"
  BLCD_LIB=$lib timeout -k 10 100 python tools/quick_bench.py Object2 200000 200 2 || exit 1
  BLCD_LIB=$lib timeout -k 10 100 python tools/quick_bench.py Bounce2 100000 200 3 || exit 1
  BLCD_LIB=$lib timeout -k 10 100 python tools/quick_bench.py Object3 100000 200 2 || exit 1
done
done
for rep in 1 2 3; do for c in 1 2; do
  echo -n "COHORTS=$c "; BLCD_COHORTS=$c timeout -k 10 100 python tools/quick_bench.py Bounce 100000 200 20 || exit 1
  echo -n "COHORTS=$c "; BLCD_COHORTS=$c timeout -k 10 100 python tools/quick_bench.py Dropbox 100000 200 20 || exit 1
done; done
