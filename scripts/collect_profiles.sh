#!/bin/bash
# One gpurun call: the round's bench line + rocprofv3 evidence.  scripts/collect_profiles.sh <round tag, e.g. r02>
# Everything lands under gpurun_out/prof_<tag>/ ; copy what should be judged into profiles/ afterwards.
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
T=$O/tiles.json
python3 $R/bench.py --steps 20 --warmup 5 --tiles-file $T > $O/bench_default.log 2>&1 || exit 1
tail -1 $O/bench_default.log > $O/bench_default.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_serial -- python3 $R/bench.py --steps 20 --warmup 3 --repeats 1 --no-cpu-baseline --no-pmc --in-flight 1 --tiles-file $T > $O/trace_serial.log 2>&1 || echo "serial trace failed"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_default -- python3 $R/bench.py --steps 20 --warmup 3 --repeats 1 --no-cpu-baseline --no-pmc --tiles-file $T > $O/trace_default.log 2>&1 || echo "default trace failed"
for pass in "sq:SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "fetch:FETCH_SIZE GRBM_GUI_ACTIVE" "write:WRITE_SIZE"; do
  n=${pass%%:*}; c=${pass#*:}
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$n -- python3 $R/bench.py --steps 4 --warmup 2 --repeats 1 --no-cpu-baseline --no-pmc --no-graph --in-flight 1 --tiles-file $T > $O/pmc_$n.log 2>&1 || echo "pmc $n failed"
done
cd $R
python3 scripts/summarize_trace.py $(ls $O/trace_serial/*/*kernel_trace.csv) 20 > $O/serial_kernel_trace_summary.md
python3 scripts/summarize_trace.py $(ls $O/trace_default/*/*kernel_trace.csv) 20 > $O/default_kernel_trace_summary.md
python3 scripts/summarize_pmc.py "Round ${tag#r}" $O/pmc_sq $O/pmc_fetch $O/pmc_write > $O/pmc_summary.md
grep "^{" $O/trace_serial.log | tail -1 > $O/bench_serial_under_rocprof.json
grep "^{" $O/trace_default.log | tail -1 > $O/bench_default_under_rocprof.json
cp $(ls $O/trace_serial/*/*kernel_stats.csv) $O/serial_rocprofv3_kernel_stats.csv
cp $(ls $O/trace_default/*/*kernel_stats.csv) $O/default_rocprofv3_kernel_stats.csv
rm -rf $O/trace_serial $O/trace_default $O/pmc_sq $O/pmc_fetch $O/pmc_write
cat $O/serial_kernel_trace_summary.md; head -40 $O/pmc_summary.md
