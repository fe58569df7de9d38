#!/bin/bash
# rocprofv3 --kernel-trace of ONE bench configuration -> gpurun_out/TAG_kernel_stats.txt   (bash tools/gpu_stats_one.sh TAG [bench args ...])
set -u; mkdir -p gpurun_out; R=$PWD; TAG=$1; shift
cd /tmp; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o bench -- python $R/bench.py --no-cpu-baseline --no-alt-precision --no-teacher-forced --no-other-configs "$@" > $R/gpurun_out/${TAG}_rocprof_bench.json 2> $R/gpurun_out/${TAG}_rocprof.err
python $R/tools/rocprof_summary.py $R/gpurun_out/prof_$TAG/bench_results.db $R/gpurun_out/${TAG}_kernel_stats.txt > /dev/null
rm -rf $R/gpurun_out/prof_$TAG; head -n 30 $R/gpurun_out/${TAG}_kernel_stats.txt | cut -c1-220
