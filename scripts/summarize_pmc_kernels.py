#!/usr/bin/env python3
"""HBM table of the kernels scripts/box_kernels_driver.py launches, from two rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE).
Usage: summarize_pmc_kernels.py <title> <pmc_fetch dir> <pmc_write dir> <spec.json>"""
import collections
import csv
import glob
import json
import re
import sys


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    m = re.match(r'([A-Za-z0-9_:]+(<[^>]*>)?)', n)
    return m.group(1) if m else n[:60]


def load(d):
    cc = sorted(glob.glob(d + '/*/*counter_collection.csv'))[-1]
    kt = sorted(glob.glob(d + '/*/*kernel_trace.csv'))[-1]
    dur = {r['Dispatch_Id']: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(kt))}
    per = collections.OrderedDict()
    for r in csv.DictReader(open(cc)):
        per.setdefault(short(r['Kernel_Name']), collections.OrderedDict()).setdefault(r['Dispatch_Id'], {})[r['Counter_Name']] = float(r['Counter_Value'])
    return per, dur


title, fdir, wdir, specf = sys.argv[1:5]
fe, dfe = load(fdir)
wr, dwr = load(wdir)
spec = json.load(open(specf))
print(f"# {title}\n")
print("`scripts/box_kernels_driver.py` under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, `--kernel-trace` for the")
print("durations); per case the LAST launch of its group (warm code objects, cold-ish data: every case allocates its own tensors).")
print("HBM MB = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 / 1e6 (gfx950 tallies wide coalesced reads at half their size: MI355X_MICROARCH.md,")
print("HBM section; narrow / scalar reads are uncalibrated there, so rows of the latency-bound kernels are upper bounds). Peak: 8 TB/s spec,")
print("6.3 TB/s achievable.\n")
print("| kernel | case | us | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM MB | achieved GB/s | algorithmic MB | x alg | algorithmic GB/s |")
print("|---|---|---:|---:|---:|---:|---:|---:|---:|---:|")
REPS = 6


def ordered(per, key):
    """every dispatch of the kernels whose name contains `key`, in launch order: (dispatch id, kernel name, counters)"""
    return sorted(((int(i), k, c) for k in per if key in k for i, c in per[k].items()), key=lambda t: t[0])


for key in spec:
    fd, wd = ordered(fe, key), ordered(wr, key)
    if not fd:
        continue
    cases = spec[key].get("cases") or [spec[key]]
    for gi, case in enumerate(cases):
        j = gi * REPS + REPS - 1                           # the last launch of the case's group
        if j >= len(fd):
            break
        i, k, c = fd[j]
        w = wd[j][2].get('WRITE_SIZE', 0.0) if j < len(wd) else 0.0
        f = c.get('FETCH_SIZE', 0.0)
        us = dfe[str(i)]
        mb = (2 * f + w) * 1024 / 1e6
        alg = case.get("algorithmic_MB")
        print(f"| `{k}` | {case.get('case', '')} | {us:.1f} | {f:.0f} | {w:.0f} | {mb:.2f} | {mb * 1e6 / (us * 1e-6) / 1e9:.0f} | "
              f"{'' if alg is None else f'{alg:.2f}'} | {'' if not alg else f'{mb / alg:.2f}'} | {'' if not alg else f'{alg * 1e6 / (us * 1e-6) / 1e9:.0f}'} |")
