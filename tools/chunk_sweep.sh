#!/bin/bash
export PYTHONPATH=/root/repo
for spec in "Bounce 100000" "Dropbox 100000"; do
  set -- $spec
  for k in "BLCD_CHUNK=30" "BLCD_CHUNK=40" "BLCD_CHUNK=50" "BLCD_CHUNK=67" "BLCD_CHUNK=100" "BLCD_CHUNK=40 BLCD_COHORTS=3" "BLCD_CHUNK=50 BLCD_COHORTS=1"; do
    echo -n "$k :: "; env $k timeout -k 10 120 python tools/quick_bench.py $1 $2 200 4 2>&1 | grep -v amdgpu | tail -1
  done
done
