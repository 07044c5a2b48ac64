#!/bin/bash
# A/B of library builds on ONE box: scripts/ab_lib.sh "<bench args>" libA.so libB.so ...   (files under two_stage_object_detection_amd/)
common=$1; shift
for v in "$@"; do
  TSOD_LIB=two_stage_object_detection_amd/$v python bench.py $common 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('[$v]', '| value', d['value'], 'serial', d['serial']['images_per_s'], 'conv seq ms', r['kernel_ms_per_forward'], 'isolated', r['kernel_ms_per_forward_isolated'], 'frac', r['frac'], 'traffic x', r.get('traffic_over_algorithmic'))"
done
