#!/bin/bash
# Round 3: does the mix gain when EVERY encoder GEMM runs on 192-row tiles (185 registers x 2 waves per SIMD leave room for
# a decode wave of <= 128 registers next to a resident GEMM workgroup; the 256-row tile's 226 do not)?
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-r03_l}; T0=$(date +%s); t() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('parity') or {}; r=d['roofline']; print(d['value'], 'captions/s', d['ms_per_step'], 'ms/pass  latency', d['batch_latency_ms']['median'], 'gemm avg us', round(r['avg_launch_ms']*1e3,1), 'enc+prefill', d['phases_ms'].get('graph_encode_prefill_ms'), 'identical', p.get('identical'))"; }
run() { # name env...
  local n=$1; shift
  local f=gpurun_out/${TAG}_bench_$n.json
  env "$@" timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 8 2>gpurun_out/${TAG}_err.txt | tail -n 1 > $f
  t "$n: $(line < $f 2>&1 | tail -n 1)"
}
for i in 1 2 3; do
  run default_$i X=1
  run all192_$i BENCH_GEMM_IMPL=16393
done
run all256_1 BENCH_GEMM_IMPL=32777
t done
