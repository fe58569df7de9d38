#!/bin/bash
# rocprofv3 kernel stats of a SOLO run (one context, eager launches): per-kernel durations comparable with the
# HIP-event numbers of bench.py's roofline pass.
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_solo -o bench -- python $R/bench.py --steps 5 --warmup 2 --contexts 1 --no-graph --no-cpu-baseline > $R/gpurun_out/rocprof_solo.txt 2>&1; echo "rc=$?"
cd $R; python tools/rocprof_summary.py gpurun_out/prof_solo/bench_results.db gpurun_out/kernel_stats_solo.txt; head -12 gpurun_out/kernel_stats_solo.txt; tail -1 gpurun_out/rocprof_solo.txt | cut -c1-900
