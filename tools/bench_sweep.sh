#!/bin/bash
# bench.py under a few BLCD_LANES / BLCD_CHUNK settings (diagnostic; run on the GPU box)
export PYTHONPATH=/root/repo
for l in 64 49; do for c in 50 20; do
  echo "lanes $l chunk $c"; BLCD_LANES=$l BLCD_CHUNK=$c timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])" || exit 1
done; done
