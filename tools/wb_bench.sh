#!/bin/bash
# in-wave batching (BLCD_WAVE_BATCH = lanes that must wait for the same kind of work before the wave resumes them) vs the plain kernel
for env in "$@"; do
  python tools/yield_bench.py --one $env 200 2>&1 | grep -v amdgpu
  for r in 4 8 16 24; do for l in 8 16 32; do
    BLCD_WAVE_BATCH=$r BLCD_YIELD_LANES=$l python tools/yield_bench.py --one $env 200 2>&1 | grep -v amdgpu | sed "s/passes=def/batch=$r/; s/lanes=def/lanes<=$l/"
  done; done
done
