#!/bin/bash
# bench.py under a few BLCD_CHUNK settings (diagnostic; run on the GPU box)
export PYTHONPATH=/root/repo
for c in 10 20 25 40 50; do
  echo -n "chunk $c: "; BLCD_CHUNK=$c timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4g env-steps/s' % d['value'], '%.3f ms/rollout' % d['ms_per_step'], 'launch %.3f ms' % d['roofline']['avg_launch_ms'])" || exit 1
done
