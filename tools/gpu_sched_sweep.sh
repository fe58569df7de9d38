#!/bin/bash
# schedule A/B on one box (bash tools/gpu_sched_sweep.sh TAG): contexts in flight x encoder chains, kernel-shape policy
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1; TAG=${1:-sweep}
F="--no-cpu-baseline --no-alt-precision --no-teacher-forced --no-other-configs --brief"
run() { n=$1; shift; timeout 200 python bench.py $F "$@" > gpurun_out/${TAG}_$n.json 2>/dev/null; python tools/bench_lines.py gpurun_out/${TAG}_$n.json | cut -c1-200; }
for rep in 1 2; do
run c4e2_$rep
run c3e1_$rep --contexts 3 --encoder-chains 1
run c4e1_$rep --contexts 4 --encoder-chains 1
run c6e2_$rep --contexts 6 --encoder-chains 2
run c6e3_$rep --contexts 6 --encoder-chains 3
run c5e2_$rep --contexts 5 --encoder-chains 2
run c4e2solo_$rep --solo-policy
run c4free_$rep --free-run
done
