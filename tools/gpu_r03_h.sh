#!/bin/bash
# Round 3, eighth GPU call: timeline of the mixed schedule (what decode launches wait for; how busy the chip is)
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD; TAG=${1:-r03_h}; T0=$(date +%s); t() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
cd /tmp; export TMPDIR=/tmp
t "trace solo"; timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_solo -o bench -- python $R/bench.py --no-cpu-baseline --contexts 1 --steps 8 --warmup 2 > /dev/null 2> $R/gpurun_out/${TAG}_solo.err; echo rc=$?
t "trace mixed"; timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_mix -o bench -- python $R/bench.py --no-cpu-baseline --steps 40 --warmup 8 > $R/gpurun_out/${TAG}_mix_bench.json 2> $R/gpurun_out/${TAG}_mix.err; echo rc=$?
cd $R
t "timeline"; python tools/mix_timeline.py /tmp/prof_mix/bench_results.db --solo /tmp/prof_solo/bench_results.db --out gpurun_out/${TAG}_mix_timeline.txt | tail -n 80
t done
