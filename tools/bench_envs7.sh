#!/bin/bash
export PYTHONPATH=/root/repo
for spec in "UrchinBalls 20000" "LuxoCubes 20000"; do
  set -- $spec
  timeout -k 10 280 python bench.py --env $1 --envs $2 --steps 1 --warmup 1 --rollout-len 100 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2', '%.4g env-steps/s' % d['value'], '%.1f ms/rollout' % d['ms_per_step'], 'faults', d['config']['faulted_envs'])" || exit 1
done
