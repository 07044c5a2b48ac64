#!/bin/bash
# A/B of bench.py options on ONE box: scripts/ab_bench.sh "<common args>" "<variant A args>" "<variant B args>" ...
common=$1; shift
for v in "$@"; do
  python bench.py $common $v 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$v', '| value', d['value'], 'serial', d['serial']['images_per_s'], 'conv seq ms', r['kernel_ms_per_forward'], 'isolated', r['kernel_ms_per_forward_isolated'], 'frac', r['frac'], 'tuning s', d['tuning_seconds']['per_rank'])"
done
