#!/bin/bash
# bench.py on the other BASELINE configs (diagnostic; run on the GPU box)
export PYTHONPATH=/root/repo
for spec in "Bounce 100000" "Dropbox 100000" "Object2 200000" "Urchin 50000" "LuxoBall 50000"; do
  set -- $spec
  timeout -k 10 280 python bench.py --env $1 --envs $2 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2', '%.4g env-steps/s' % d['value'], '%.1f ms/rollout' % d['ms_per_step'], 'faults', d['config']['faulted_envs'])" || exit 1
done
