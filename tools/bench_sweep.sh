#!/bin/bash
# bench.py under a few BLCD_CHUNK settings (diagnostic; run on the GPU box)
for c in 50 25 10; do
  echo "chunk $c"; BLCD_CHUNK=$c timeout -k 10 200 python bench.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])" || exit 1
done
