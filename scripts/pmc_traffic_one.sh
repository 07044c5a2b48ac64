#!/bin/bash
# HBM traffic and L2 hit rate of one conv configuration: scripts/pmc_traffic_one.sh <tag> B H W Cin Cout k tile split prec
# (separate --pmc passes: FETCH_SIZE, WRITE_SIZE, TCC_HIT_sum TCC_MISS_sum; bytes = (2 FETCH + WRITE) * 1024: the guide's gfx950 correction)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/pmct_${tag}_$n -- python3 $R/scripts/conv_one.py $1 $2 $3 $4 $5 $6 $7 $8 6 $9 > /dev/null 2>&1 || echo "pass $n failed"
done
python3 - <<PY
import csv, glob, collections
vals = {}
for d in sorted(glob.glob("$R/gpurun_out/pmct_${tag}_*")):
    f = glob.glob(d + "/*/*counter_collection.csv")
    if not f: continue
    kt = glob.glob(d + "/*/*kernel_trace.csv")[0]
    dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(kt))}
    acc = collections.defaultdict(list); us = []
    for r in csv.DictReader(open(f[0])):
        if "conv_igemm" in r["Kernel_Name"] or "conv_dma" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"])); us.append(dur[r["Dispatch_Id"]])
    for k, v in acc.items():
        vals[k] = sum(v[1:]) / max(1, len(v) - 1)
    vals["us"] = sum(us[1:]) / max(1, len(us) - 1)
mb = (2 * vals.get("FETCH_SIZE", 0) + vals.get("WRITE_SIZE", 0)) * 1024 / 1e6
h, m = vals.get("TCC_HIT_sum", 0), vals.get("TCC_MISS_sum", 0)
print(f"$tag: {vals.get('us', 0):.1f} us  HBM {mb:.1f} MB (fetch {2 * vals.get('FETCH_SIZE', 0) * 1024 / 1e6:.1f}, write {vals.get('WRITE_SIZE', 0) * 1024 / 1e6:.1f})  L2 hit {100 * h / max(1, h + m):.0f}%")
PY
rm -rf $R/gpurun_out/pmct_${tag}_*
