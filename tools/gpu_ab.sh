#!/bin/bash
# Interleaved A/B of whole bench.py runs on ONE box (GPU boxes differ by ~5 %: only pairs taken on the same box compare).
#   tools/gpu_ab.sh TAG ROUNDS "name1:ENV1=v ENV2=v -- extra bench args" "name2:..." ...
# Every variant runs through libgitmi_exp.so (bench.py --experiment), whose engine reads the GITMI_* overrides; variants
# are run round-robin ROUNDS times; one summary line per run (tools/bench_lines.py), JSON lines under gpurun_out/TAG_*.
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=$1; ROUNDS=$2; shift 2
T0=$(date +%s); t() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
STEPS=${AB_STEPS:-40}; WARM=${AB_WARMUP:-8}
for r in $(seq 1 $ROUNDS); do
  for spec in "$@"; do
    name=${spec%%:*}; rest=${spec#*:}
    envs=${rest%%--*}; extra=""
    case "$rest" in *--*) extra=${rest#*--};; esac
    f=gpurun_out/${TAG}_${name}_$r.json
    env $envs timeout ${AB_TIMEOUT:-300} python bench.py --experiment --no-cpu-baseline --steps $STEPS --warmup $WARM $extra \
        2> gpurun_out/${TAG}_${name}_$r.err | tail -n 1 > $f
    t "$(python tools/bench_lines.py $f | cut -c1-200)"
    [ -s $f ] || tail -n 5 gpurun_out/${TAG}_${name}_$r.err
  done
done
t done
