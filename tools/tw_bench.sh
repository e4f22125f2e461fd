#!/bin/bash
# two wave widths (awake slots in narrower waves once they no longer fill the SIMDs): narrowest width 0 (off) / 16 / 32 / 48
for env in "$@"; do
  for tw in 0 8 16 24; do
    BLCD_TWO_WIDTHS=$tw python tools/yield_bench.py --one $env 200 2>&1 | grep -v amdgpu | sed "s/passes=def/two_widths=$tw/"
  done
done
