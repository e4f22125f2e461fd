#!/bin/bash
# throughput of asynchronous rollouts (BLCD_ASYNC = world steps per launch) against the plain fused rollout
for env in "$@"; do
  python tools/yield_bench.py --one $env 200 2>&1 | grep -v amdgpu
  for k in 1 2 3 6; do for l in 8 24; do
    BLCD_ASYNC=$k BLCD_YIELD_LANES=$l python tools/yield_bench.py --one $env 200 2>&1 | grep -v amdgpu | sed "s/passes=def/async=$k/; s/lanes=def/lanes<=$l/"
  done; done
done
