#!/bin/bash
# copy what scripts/collect_profiles.sh left under gpurun_out/prof_<tag>_<name>/ into profiles/<tag>_<name>_*
tag=$1; n=$2
O=gpurun_out/prof_${tag}_$n
cp $O/bench_default.json profiles/${tag}_${n}_bench_default.json
cp $O/serial_kernel_trace_summary.md profiles/${tag}_${n}_serial_kernel_trace_summary.md
cp $O/serial_rocprofv3_kernel_stats.csv profiles/${tag}_${n}_serial_rocprofv3_kernel_stats.csv
cp $O/bench_serial_under_rocprof.json profiles/${tag}_${n}_bench_serial_under_rocprof.json
cp $O/pmc_summary.md profiles/${tag}_${n}_pmc_summary.md
cp $O/tiles.json profiles/${tag}_${n}_autotuned_tiles.json
cp $O/layers.json profiles/${tag}_${n}_layers.json
if [ -f $O/default_kernel_trace_summary.md ]; then
  cp $O/default_kernel_trace_summary.md profiles/${tag}_${n}_inflight4_kernel_trace_summary.md
  cp $O/default_rocprofv3_kernel_stats.csv profiles/${tag}_${n}_inflight4_rocprofv3_kernel_stats.csv
  cp $O/bench_default_under_rocprof.json profiles/${tag}_${n}_bench_inflight4_under_rocprof.json
fi
