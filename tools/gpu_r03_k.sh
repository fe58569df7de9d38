#!/bin/bash
# Round 3: how far does sharing one decode chain between several 64-image requests go?  Sweep of requests per engine pass
# (--coalesce) x contexts in flight x encoders at a time; every line carries batch_latency_ms and timed_ids_equal_solo.
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-r03_k}; T0=$(date +%s); t() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('parity') or {}; print(d['value'], 'captions/s', d['ms_per_step'], 'ms/pass  latency', d['batch_latency_ms'], 'ids==solo', d.get('timed_ids_equal_solo'), 'identical', p.get('identical'))"; }
run() { # coalesce contexts chains
  local f=gpurun_out/${TAG}_bench_co$1_c$2_e$3.json
  timeout 300 python bench.py --no-cpu-baseline --steps 48 --warmup 8 --coalesce $1 --contexts $2 --encoder-chains $3 2>gpurun_out/${TAG}_err.txt | tail -n 1 > $f
  t "coalesce $1 contexts $2 chains $3: $(line < $f 2>&1 | tail -n 1)"
}
run 1 4 2
run 2 2 1
run 2 2 2
run 2 3 1
run 2 3 2
run 2 4 2
run 4 2 1
run 4 2 2
run 4 3 2
run 3 3 2
run 1 4 2
t done
