#!/bin/bash
# Serial-schedule kernel trace of the B=1 forward (tiles from an earlier run): scripts/quick_trace.sh <tiles.json> <out name> [bench args]
T=$1; N=$2; shift 2
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/qt_$N
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 20 --warmup 3 --repeats 1 --no-cpu-baseline --no-pmc --in-flight 1 --tiles-file $R/$T "$@" > $O/trace.log 2>&1 || echo "trace failed"
cd $R
python3 scripts/summarize_trace.py $(ls $O/trace/*/*kernel_trace.csv) 20 > $O/summary.md
rm -rf $O/trace
cat $O/summary.md
