#!/bin/bash
# scripts/collect_box_profiles.sh <round tag>: PMC passes over scripts/box_kernels_driver.py -> gpurun_out/prof_<tag>_box/
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_${tag}_box
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  n=${pass%%:*}; c=${pass#*:}
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$n -- python3 $R/scripts/box_kernels_driver.py $O/spec.json > $O/pmc_$n.log 2>&1 || echo "pmc $n failed"
done
cd $R
python3 scripts/summarize_pmc_kernels.py "Round ${tag#r} - HBM traffic of the box / depthwise / layout kernels at full size" $O/pmc_fetch $O/pmc_write $O/spec.json > $O/box_kernels_pmc_summary.md
rm -rf $O/pmc_fetch $O/pmc_write
cat $O/box_kernels_pmc_summary.md
