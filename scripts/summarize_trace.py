#!/usr/bin/env python3
"""Per-forward kernel summary (markdown) from a rocprofv3 --kernel-trace CSV of a bench.py run.
Usage: summarize_trace.py <kernel_trace.csv> <n_last_forwards> [counter_collection.csv ...]"""
import collections, csv, re, sys

def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    m = re.match(r'([A-Za-z0-9_:]+(<[^>]*>)?)', n)
    return m.group(1) if m else n[:60]

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = int(sys.argv[2])
# the first kernel of every forward: the one-launch stem, or the layout kernel where the stem is three launches
opener = 'stem_kernel' if any('stem_kernel' in r['Kernel_Name'] for r in rows[-200:]) else 'nchw_to_nhwc_small'
marks = [i for i, r in enumerate(rows) if opener in r['Kernel_Name']]
sel = rows[marks[-n]:]
per = collections.OrderedDict()
for r in sel:
    k = short(r['Kernel_Name'])
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    e = per.setdefault(k, [0, 0.0, 1e30, 0.0])
    e[0] += 1; e[1] += d; e[2] = min(e[2], d); e[3] = max(e[3], d)
span = (int(sel[-1]['End_Timestamp']) - int(sel[0]['Start_Timestamp'])) / 1e3 / n
tot = sum(v[1] for v in per.values()) / n
print(f"steady state: last {n} forwards (HIP-graph replays); kernel time {tot:.1f} us/forward, wall span {span:.1f} us/forward\n")
print("| kernel | launches/forward | us/forward | avg us | min us | max us | share |")
print("|---|---:|---:|---:|---:|---:|---:|")
for k, v in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print(f"| `{k}` | {v[0] / n:.1f} | {v[1] / n:.1f} | {v[1] / v[0]:.2f} | {v[2]:.2f} | {v[3]:.2f} | {100 * v[1] / n / tot:.1f}% |")
