#!/bin/bash
# counters of one conv configuration: scripts/pmc_one.sh <tag> B H W Cin Cout k tile split prec
# three separate --pmc passes (SQ busy / wait, LDS, instruction mix); results under gpurun_out/pmc_<tag>_*/
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
            "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS" \
            "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA"; do
  n=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_$n -- python3 $R/scripts/conv_one.py $1 $2 $3 $4 $5 $6 $7 $8 6 $9 > /dev/null 2>&1 || echo "pass $n failed"
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$R/gpurun_out/pmc_${tag}_*")):
    f = glob.glob(d + "/*/*counter_collection.csv")
    if not f: continue
    kt = glob.glob(d + "/*/*kernel_trace.csv")[0]
    dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(kt))}
    acc = collections.defaultdict(list); us = []
    for r in csv.DictReader(open(f[0])):
        if "conv_igemm" in r["Kernel_Name"] or "conv_dma" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"])); us.append(dur[r["Dispatch_Id"]])
    if not us: continue
    t = sum(us) / len(us)
    print(f"{d.split('/')[-1]}: avg {t:.1f} us; " + "  ".join(f"{k}={sum(v[1:]) / max(1, len(v) - 1):.3g}" for k, v in acc.items()))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in acc:
        v = acc["SQ_VALU_MFMA_BUSY_CYCLES"]; m = sum(v[1:]) / (len(v) - 1)
        print(f"   MFMA busy = {100 * m / (4 * 256 * t * 1e-6 * 2.4e9):.1f}% of SIMD-cycles at 2.4 GHz")
PY
