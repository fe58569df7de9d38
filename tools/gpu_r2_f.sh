#!/bin/bash
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
R=$PWD
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_f -o bench -- python $R/bench.py --steps 6 --warmup 2 --contexts 1 --no-cpu-baseline > $R/gpurun_out/f_rocprof_solo.txt 2>&1; echo "rocprof rc=$?"
cd $R
python tools/rocprof_summary.py gpurun_out/prof_f/bench_results.db gpurun_out/f_kernel_stats_solo_graph.txt > /dev/null; head -n 22 gpurun_out/f_kernel_stats_solo_graph.txt | cut -c1-200
rm -rf gpurun_out/prof_f
