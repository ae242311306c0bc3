#!/bin/bash
# weight-gradient micro-benchmark: experiment variants of the 160-tile kernel
for w in 0 1 2 3; do
  echo "PSVAE_WD_EXP=$w"
  PSVAE_WD_EXP=$w timeout -k 10 120 python3 tools/dw_bench.py 95000x300x300 || exit 1
done
