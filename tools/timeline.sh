#!/bin/bash
# GPU box: per-dispatch timeline (start, duration, kernel, grid) of a few rollouts of one workload
# usage: tools/timeline.sh <tag> <Env> <envs> [rollouts]   -> gpurun_out/<tag>_timeline.txt (+ the kernel trace CSV)
export PYTHONPATH=/root/repo
REPO=$PWD
TAG=$1; ENV=$2; N=$3; R=${4:-2}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl_$TAG
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/tl_$TAG -o tl --output-format csv -- python3 $REPO/tools/quick_bench.py $ENV $N 200 $R > $REPO/gpurun_out/${TAG}_timeline_run.log 2>&1 || { tail -5 $REPO/gpurun_out/${TAG}_timeline_run.log; exit 1; }
f=$(find /tmp/tl_$TAG -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY' > $REPO/gpurun_out/${TAG}_timeline.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last rollout only: from the last reset_kernel on
starts = [i for i, r in enumerate(rows) if 'reset_kernel' in r['Kernel_Name']]
rows = rows[starts[-1]:] if starts else rows
t0 = int(rows[0]['Start_Timestamp'])
print('start_ms  dur_ms  queue  grid  kernel')
busy = {}
for r in rows:
  s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
  k = r['Kernel_Name'].split('(')[0].replace('void blcd::', '')[:60]
  print('%9.3f %8.3f %5s %8s  %s' % ((s - t0) / 1e6, (e - s) / 1e6, r.get('Queue_Id', '?'), r.get('Grid_Size', '?'), k))
  busy[k] = busy.get(k, 0) + (e - s)
print('total span ms', (max(int(r['End_Timestamp']) for r in rows) - t0) / 1e6)
for k, v in sorted(busy.items(), key=lambda x: -x[1]):
  print('  sum %-60s %.3f ms' % (k, v / 1e6))
PY
tail -12 $REPO/gpurun_out/${TAG}_timeline.txt
cat $REPO/gpurun_out/${TAG}_timeline_run.log | tail -2
