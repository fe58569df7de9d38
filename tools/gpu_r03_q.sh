#!/bin/bash
# Round 3: what does each kernel class of the decode chain cost the MIXED schedule?  The launches of one class at a time are
# left out of the chain (GITMI_DECODE_SKIP; ids are garbage, the encoders and everything else run unchanged).
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-r03_q}; T0=$(date +%s); t() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline_decode']; print(d['value'], 'captions/s', d['ms_per_step'], 'ms/pass  latency', d['batch_latency_ms']['median'], 'solo dec step', r['avg_step_ms'], 'solo enc', d['phases_ms']['graph_encode_prefill_ms'])"; }
run() { # name skip
  local f=gpurun_out/${TAG}_bench_$1.json
  GITMI_DECODE_SKIP=$2 timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 8 2>gpurun_out/${TAG}_err_$1.txt | tail -n 1 > $f
  t "skip $2 ($1): $(line < $f 2>&1 | tail -n 1)"
}
run none_1 0
run attention 1
run gemm_qkv_ffn1 2
run gemm_out_ffn2 4
run vocab 8
run all_gemms 6
run attention_gemms 7
run all 15
run none_2 0
t done
