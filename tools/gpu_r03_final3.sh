#!/bin/bash
# Round 3, last call: PMC passes for the FINAL csrc (copied to profiles/ first, so the bench line of this run carries fresh
# `traffic`), the default line again, and a sweep of contexts in flight x encoders at a time under the serving policy
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-r03_zz}; T0=$(date +%s); t() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
t "pmc"; bash tools/gpu_pmc.sh > gpurun_out/${TAG}_pmc.log 2>&1; cp gpurun_out/pmc_summary.tsv gpurun_out/${TAG}_pmc_summary.tsv; cp gpurun_out/pmc_summary.tsv profiles/${TAG}_pmc_summary.tsv; rm -rf gpurun_out/pmc; head -n 5 gpurun_out/${TAG}_pmc_summary.tsv | cut -c1-160
t "default bench"; timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "rc=$?"; python -c "import json; d=json.loads(open('gpurun_out/${TAG}_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source'], d['roofline']['traffic_stale'], d['roofline_decode']['traffic'], d['cpu_baseline']['value'])"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('parity') or {}; print(d['value'], 'captions/s', d['ms_per_step'], 'ms/pass  latency', d['batch_latency_ms']['median'], 'ids==solo', d.get('timed_ids_equal_solo'), 'identical', p.get('identical'))"; }
run() { local n=$1; shift; local f=gpurun_out/${TAG}_sweep_$n.json; timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 8 "$@" 2>/dev/null | tail -n 1 > $f; t "$n: $(line < $f 2>&1 | tail -n 1)"; }
run c4_e2 --contexts 4 --encoder-chains 2
run c5_e2 --contexts 5 --encoder-chains 2
run c6_e2 --contexts 6 --encoder-chains 2
run c6_e3 --contexts 6 --encoder-chains 3
run c8_e2 --contexts 8 --encoder-chains 2
run c8_e4 --contexts 8 --encoder-chains 4
run c3_e1 --contexts 3 --encoder-chains 1
run c4_free --contexts 4 --free-run
run c4_e2_b --contexts 4 --encoder-chains 2
t done
