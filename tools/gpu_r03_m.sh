#!/bin/bash
# Round 3: confirm r03_l's single sample -- 256-row tiles for EVERY encoder GEMM (N = 768 launches: 150 instead of 198
# workgroups) in the mixed schedule, interleaved with the default; solo figures next to it
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-r03_m}; T0=$(date +%s); t() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('parity') or {}; r=d['roofline']; print(d['value'], 'captions/s', d['ms_per_step'], 'ms/pass  latency', d['batch_latency_ms']['median'], 'gemm avg us', round(r['avg_launch_ms']*1e3,1), 'enc+prefill', d['phases_ms'].get('graph_encode_prefill_ms'), 'identical', p.get('identical'))"; }
run() { # name args... (env via BENCH_GEMM_IMPL exported by caller)
  local n=$1; shift
  local f=gpurun_out/${TAG}_bench_$n.json
  timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 8 "$@" 2>gpurun_out/${TAG}_err.txt | tail -n 1 > $f
  t "$n: $(line < $f 2>&1 | tail -n 1)"
}
for i in 1 2 3; do
  unset BENCH_GEMM_IMPL; run default_$i
  export BENCH_GEMM_IMPL=32777; run all256_$i
done
unset BENCH_GEMM_IMPL; run default_c1 --contexts 1
export BENCH_GEMM_IMPL=32777; run all256_c1 --contexts 1
unset BENCH_GEMM_IMPL; run default_c8 --contexts 8
export BENCH_GEMM_IMPL=32777; run all256_c8 --contexts 8
export BENCH_GEMM_IMPL=32777; run all256_c6e3 --contexts 6 --encoder-chains 3
export BENCH_GEMM_IMPL=32777; run all256_c4e1 --contexts 4 --encoder-chains 1
export BENCH_GEMM_IMPL=32777; run all256_c4free --contexts 4 --free-run
t done
