#!/bin/bash
# Round 3: bf16 / fp16 GEMM outputs stored write-through WITHOUT keeping the line in the XCD's L2 (sc1): does the 58-78 MB
# output stream of a launch stop evicting the weight panels?  Isolated launches + the mixed bench, interleaved.
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-r03_r}; T0=$(date +%s); t() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('parity') or {}; r=d['roofline']; print(d['value'], 'captions/s', d['ms_per_step'], 'ms/pass', 'gemm avg us', round(r['avg_launch_ms']*1e3,1), 'frac', r['frac'], 'enc+prefill', d['phases_ms'].get('graph_encode_prefill_ms'), 'identical', p.get('identical'))"; }
run() { # name impl
  local f=gpurun_out/${TAG}_bench_$1.json
  if [ "$2" = "-" ]; then unset BENCH_GEMM_IMPL; else export BENCH_GEMM_IMPL=$2; fi
  timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 8 2>gpurun_out/${TAG}_err_$1.txt | tail -n 1 > $f
  t "$1: $(line < $f 2>&1 | tail -n 1)"
}
for i in 1 2 3; do
  run default_$i -
  run sc1_$i 131081
done
run all256_1 32777
run all256_sc1_1 163849
run all256_2 32777
run all256_sc1_2 163849
t done
