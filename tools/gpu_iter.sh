#!/bin/bash
# Iteration run: op tests, parity, bench, rocprof stats -> gpurun_out/
set -u
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
TESTS="${1:-tests}"
echo "== pytest"; timeout 900 python -m pytest $TESTS -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.txt 2>&1; echo "rc=$?"; tail -15 gpurun_out/pytest_gpu.txt
echo "== bench"; timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench.txt 2>&1; echo "rc=$?"; tail -1 gpurun_out/bench.txt
echo "== rocprof"; R=$PWD; cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/rocprof.txt 2>&1; echo "rc=$?"; cd $R
python tools/rocprof_summary.py gpurun_out/prof/bench_results.db gpurun_out/kernel_stats.txt; head -25 gpurun_out/kernel_stats.txt
