#!/bin/bash
# Short round-end collection (when only a few GPU-minutes are left), most valuable first:
#   1. rocprofv3 --pmc passes for the CURRENT csrc -> profiles/<tag>_pmc_summary.tsv (so that bench.py's traffic is not stale)
#   2. the default bench line under rocprofv3 --kernel-trace (+ per-kernel stats of the same command)
#   3. the default bench line without a profiler
#   4. a slice of the parity suite (the BASELINE-shape goldens)
# usage: bash tools/gpu_round_end_short.sh <tag>
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD; TAG=${1:-short}; T0=$(date +%s)
stamp() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
stamp pmc; bash tools/gpu_pmc.sh > gpurun_out/${TAG}_pmc.log 2>&1
cp gpurun_out/pmc_summary.tsv gpurun_out/${TAG}_pmc_summary.tsv && cp gpurun_out/pmc_summary.tsv profiles/${TAG}_pmc_summary.tsv
rm -rf gpurun_out/pmc; head -n 3 gpurun_out/${TAG}_pmc_summary.tsv | cut -c1-120
stamp rocprof bench
( cd /tmp; export TMPDIR=/tmp
  timeout 150 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_default -o bench -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/${TAG}_default_bench.json 2> $R/gpurun_out/${TAG}_default.err
  python $R/tools/rocprof_summary.py $R/gpurun_out/prof_default/bench_results.db $R/gpurun_out/${TAG}_default_kernel_stats.txt > /dev/null
  rm -rf $R/gpurun_out/prof_default )
tail -n 1 gpurun_out/${TAG}_default_bench.json | cut -c1-400
stamp plain bench
timeout 120 python bench.py --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -n 1 gpurun_out/${TAG}_bench.json | cut -c1-1500
stamp parity slice
timeout 200 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider --tb=short -k "full_bench or full_base or base_greedy" 2>&1 | tail -n 4 | tee gpurun_out/${TAG}_parity_slice.txt
stamp done
