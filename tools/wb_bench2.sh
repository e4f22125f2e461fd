#!/bin/bash
for env in "$@"; do
  python tools/yield_bench.py --one $env 200 2>&1 | grep -v amdgpu
  for r in 2 6 16; do for l in 8 32; do
    BLCD_WAVE_BATCH=$r BLCD_YIELD_LANES=$l python tools/yield_bench.py --one $env 200 2>&1 | grep -v amdgpu | sed "s/passes=def/batch=$r/; s/lanes=def/lanes<=$l/"
  done; done
done
