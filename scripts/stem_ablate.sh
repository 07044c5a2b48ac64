#!/bin/bash
# Where a stem tile's time goes: scripts/stem_bench.py with phases of stem_kernel switched off (TSOD_STEM_DBG, wrong results by design).
for d in 0 1 2 4 8 6 7 14 15; do
  echo "TSOD_STEM_DBG=$d"
  TSOD_STEM_DBG=$d timeout -k 10 100 python scripts/stem_bench.py 1 8 2>&1 | grep "B=" | cut -c1-90
done
