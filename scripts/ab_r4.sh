#!/bin/bash
# same-box A/B of this round's switches: scripts/ab_r4.sh <out dir under gpurun_out> [bench args]
O=gpurun_out/$1; shift
mkdir -p $O
run() { n=$1; shift; "$@" > $O/$n.out 2> $O/$n.err; tail -1 $O/$n.out > $O/$n.json; python3 - $O/$n.json $n <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print(f"{sys.argv[2]:28s} in flight {d['value']:8.1f} img/s  serial {d['serial']['ms_per_step']:.4f} ms  convs {r['kernel_ms_per_forward']:.4f} ms  frac {r['frac']:.3f}  "
      f"fused {d.get('roofline', {}).get('kernel', '').count('bottleneck_kernel')}  tune {d['tuning_seconds']['per_rank']}  parity {d.get('parity', {}).get('ok') if d.get('parity') else None}", flush=True)
PY
}
run default python bench.py --no-pmc --cpu-reps 2 "$@"
run fuse_off python bench.py --no-pmc --cpu-reps 2 --fuse-bottleneck off "$@"
run no_range_words env TSOD_NO_RANGE_WORDS=1 python bench.py --no-pmc --cpu-reps 2 --fuse-bottleneck off "$@"
run default_again python bench.py --no-pmc --cpu-reps 2 "$@"
