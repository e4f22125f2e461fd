export PYTHONPATH=/root/repo
for sp in 500 520 540 560 580 460; do
  echo -n "SPLIT=$sp :: "; BLCD_COHORT_SPLIT=$sp timeout -k 10 120 python tools/quick_bench.py Bounce 100000 200 5 || exit 1
done
for sp in 500 520 540 560; do
  echo -n "SPLIT=$sp :: "; BLCD_COHORT_SPLIT=$sp timeout -k 10 120 python tools/quick_bench.py Dropbox 100000 200 5 || exit 1
done
for sp in 500 530 560; do
  echo -n "SPLIT=$sp :: "; BLCD_COHORT_SPLIT=$sp timeout -k 10 200 python tools/quick_bench.py Object2 200000 200 2 || exit 1
done
