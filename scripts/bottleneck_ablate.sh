#!/bin/bash
# timing-only ablations of bottleneck_kernel (TSOD_BN_DBG bits: 1 no x loads, 2 no residual loads, 4 no weight loads, 8 no stores)
for d in 0 1 2 4 8 3 7 15; do echo "TSOD_BN_DBG=$d"; TSOD_BN_DBG=$d python scripts/bottleneck_bench.py 1 8 2>&1 | grep "B="; done
