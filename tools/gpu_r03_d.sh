#!/bin/bash
# Round 3, fourth GPU call: the p8 schedule variants (10-slot ring; ring + two merged phases per K tile)
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-r03_d}; T0=$(date +%s); t() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
t "p8 variant unit tests"; timeout 900 python -m pytest tests/test_gpu_ops.py -q --tb=short -p no:cacheprovider -x -k "p8_variant or test_gemm_bf16" > gpurun_out/${TAG}_pytest_p8.txt 2>&1; echo "rc=$?"; tail -n 6 gpurun_out/${TAG}_pytest_p8.txt | cut -c1-250
t "isolated GEMM A/B"; timeout 600 python tools/gemm_p9_bench.py 2>&1 | grep -v amdgpu | tee gpurun_out/${TAG}_gemm_sched_bench.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('parity') or {}; print(d['value'], 'captions/s', d['ms_per_step'], 'ms/pass | gemm frac', d['roofline']['frac'], 'avg launch ms', d['roofline']['avg_launch_ms'], '| enc+prefill', d['phases_ms']['graph_encode_prefill_ms'], 'decode', d['phases_ms']['graph_decode_ms'], '| parity', p.get('identical'), p.get('ok'), p.get('logit_err'))"; }
for i in 1 2; do
  for sc in 0 1 2; do
    t "bench P8_SCHED=$sc ($i)"; GITMI_P8_SCHED=$sc timeout 600 python bench.py --no-cpu-baseline --steps 40 --warmup 8 2> gpurun_out/${TAG}_sched.err | tee gpurun_out/${TAG}_bench_sched${sc}_$i.json | line
  done
done
t done
