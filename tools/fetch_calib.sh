#!/bin/bash
# GPU box: FETCH_SIZE / WRITE_SIZE of kernels with known byte counts (tools/micro/fetch_calib.hip) -> gpurun_out/fetch_calib.txt
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fc_*
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 120 rocprofv3 --pmc $c --kernel-trace -d /tmp/fc_$c -o fc --output-format csv -- $REPO/tools/micro/fetch_calib > /tmp/fc_$c.log 2>&1 || { tail -5 /tmp/fc_$c.log; exit 1; }
done
python3 - <<PY > $REPO/gpurun_out/fetch_calib.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob('/tmp/fc_*/**/*counter_collection.csv', recursive=True):
  for r in csv.DictReader(open(f)):
    acc[(r['Kernel_Name'].split('(')[0], r['Counter_Name'])].append(float(r['Counter_Value']))
B = float(1 << 30)
print('kernel counter mean_KB counter_bytes/true_bytes (true = 1 GiB per kernel)')
for (k, c), v in sorted(acc.items()):
  m = sum(v) / len(v)
  print(k, c, '%.1f' % m, '%.4f' % (m * 1024.0 / B))
PY
cat $REPO/gpurun_out/fetch_calib.txt
