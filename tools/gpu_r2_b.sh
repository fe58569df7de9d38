#!/bin/bash
# round 2: full GPU suite, default bench line, solo graph-replay kernel trace
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s > gpurun_out/b_pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 6 gpurun_out/b_pytest.txt
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/b_bench.txt 2>&1; echo "bench rc=$?"; tail -n 1 gpurun_out/b_bench.txt | cut -c1-300
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_solo_graph -o bench -- python $R/bench.py --steps 6 --warmup 2 --contexts 1 --no-cpu-baseline > $R/gpurun_out/b_rocprof_solo.txt 2>&1; echo "rocprof rc=$?"
cd $R
python tools/rocprof_summary.py gpurun_out/prof_solo_graph/bench_results.db gpurun_out/b_kernel_stats_solo_graph.txt > /dev/null; head -n 30 gpurun_out/b_kernel_stats_solo_graph.txt | cut -c1-220
rm -rf gpurun_out/prof_solo_graph/*.csv
