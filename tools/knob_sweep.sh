#!/bin/bash
# GPU box: quick_bench over a few run-time knobs for the re-binned (joint-free) BASELINE batches; one line per setting.
export PYTHONPATH=/root/repo
for spec in "Bounce 100000" "Dropbox 100000" "Object2 200000"; do
  set -- $spec
  for k in "BLCD_COHORTS=2" "BLCD_COHORTS=3" "BLCD_COHORTS=1" "BLCD_CHUNK=10" "BLCD_CHUNK=40" "BLCD_TWO_WIDTHS=1" "BLCD_TWO_WIDTHS=0"; do
    echo -n "$k :: "; env $k timeout -k 10 120 python tools/quick_bench.py $1 $2 200 3 2>&1 | grep -v amdgpu | tail -1
  done
done
