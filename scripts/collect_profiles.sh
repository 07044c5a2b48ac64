#!/bin/bash
# One gpurun call: a workload's bench line + rocprofv3 evidence.
#   scripts/collect_profiles.sh <round tag, e.g. r03> <name, e.g. b1> [bench.py workload args, e.g. --batch 8]
# Everything lands under gpurun_out/prof_<tag>_<name>/ ; copy what should be judged into profiles/ afterwards.
# The first bench run autotunes and writes the tile tables + head-GEMM choices; every profiled run re-uses that file, so no
# profiled process contains tuning launches.  Under rocprofv3 the program itself (python3) follows `--`.
tag=$1; name=$2; shift 2
R=$GRAFT_REPO_ROOT
# batch / backbone of the workload (for the summary's algorithmic-bytes column)
BATCH=1; BACKBONE=resnet50; prev=""
for a in "$@"; do
  [ "$prev" = "--batch" ] && BATCH=$a
  [ "$prev" = "--backbone" ] && BACKBONE=$a
  prev=$a
done
O=$R/gpurun_out/prof_${tag}_$name
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
T=$O/tiles.json
python3 $R/bench.py --steps 20 --warmup 5 --tiles-file $T --dump-layers $O/layers.json "$@" > $O/bench_default.log 2>&1 || { tail -20 $O/bench_default.log; exit 1; }
tail -1 $O/bench_default.log > $O/bench_default.json
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_serial -- python3 $R/bench.py --steps 20 --warmup 3 --repeats 1 --no-cpu-baseline --no-pmc --in-flight 1 --tiles-file $T "$@" > $O/trace_serial.log 2>&1 || echo "serial trace failed"
echo "serial trace done"
if [ "$name" = "b1" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_default -- python3 $R/bench.py --steps 20 --warmup 3 --repeats 1 --no-cpu-baseline --no-pmc --tiles-file $T "$@" > $O/trace_default.log 2>&1 || echo "default trace failed"
echo "default trace done"
fi
for pass in "sq:SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "fetch:FETCH_SIZE GRBM_GUI_ACTIVE" "write:WRITE_SIZE" "tcc:TCC_HIT_sum TCC_MISS_sum"; do
  n=${pass%%:*}; c=${pass#*:}
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$n -- python3 $R/bench.py --steps 4 --warmup 2 --repeats 1 --no-cpu-baseline --no-pmc --no-graph --in-flight 1 --tiles-file $T "$@" > $O/pmc_$n.log 2>&1 || echo "pmc $n failed"
  echo "pmc $n done"
done
cd $R
python3 scripts/summarize_trace.py $(ls $O/trace_serial/*/*kernel_trace.csv) 20 > $O/serial_kernel_trace_summary.md
python3 scripts/summarize_pmc.py "Round ${tag#r} ($name)" $O/pmc_sq $O/pmc_fetch $O/pmc_write --layers $O/layers.json --tcc $O/pmc_tcc --workload "$name: bench.py $*" --batch $BATCH --backbone $BACKBONE > $O/pmc_summary.md
grep "^{" $O/trace_serial.log | tail -1 > $O/bench_serial_under_rocprof.json
cp $(ls $O/trace_serial/*/*kernel_stats.csv) $O/serial_rocprofv3_kernel_stats.csv
if [ "$name" = "b1" ]; then
python3 scripts/summarize_trace.py $(ls $O/trace_default/*/*kernel_trace.csv) 20 > $O/default_kernel_trace_summary.md
grep "^{" $O/trace_default.log | tail -1 > $O/bench_default_under_rocprof.json
cp $(ls $O/trace_default/*/*kernel_stats.csv) $O/default_rocprofv3_kernel_stats.csv
fi
rm -rf $O/trace_serial $O/trace_default $O/pmc_sq $O/pmc_fetch $O/pmc_write $O/pmc_tcc
cat $O/serial_kernel_trace_summary.md; tail -70 $O/pmc_summary.md
