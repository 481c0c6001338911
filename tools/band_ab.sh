#!/bin/bash
# LU reuse band A/B on the C3 solves: wall + statistics (tools/solve_stats.py) and deviation from the truths (tools/config_stats.py)
for b in 0.35 0.40 0.45 0.50; do
  echo "band $b"
  KIN_LU_BAND=$b python tools/solve_stats.py 10000 50000 100
  KIN_LU_BAND=$b python tools/config_stats.py mid long c4long 2>/dev/null | grep -E "chunkwise|c4_long_default"
done
