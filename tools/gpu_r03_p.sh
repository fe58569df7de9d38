#!/bin/bash
# Round 3: decode attention packed onto fewer CUs (GITMI_ATTN_PW = (sentence, head) pairs per workgroup) in the mixed schedule
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-r03_p}; T0=$(date +%s); t() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
t "attention tests"; timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_group.py -q --tb=short -p no:cacheprovider -x -k "attn_decode or group" 2>&1 | tail -n 3
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('parity') or {}; r=d['roofline_decode']; print(d['value'], 'captions/s', d['ms_per_step'], 'ms/pass  latency', d['batch_latency_ms']['median'], 'dec step', r['avg_step_ms'], 'ids==solo', d.get('timed_ids_equal_solo'), 'identical', p.get('identical'))"; }
run() { # name args...
  local n=$1; shift
  local f=gpurun_out/${TAG}_bench_$n.json
  timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 8 "$@" 2>gpurun_out/${TAG}_err_$n.txt | tail -n 1 > $f
  t "$n: $(line < $f 2>&1 | tail -n 1)"
  [ -s $f ] || tail -n 5 gpurun_out/${TAG}_err_$n.txt
}
for i in 1 2 3; do
  unset GITMI_ATTN_PW; run pw1_$i
  export GITMI_ATTN_PW=4; run pw4_$i
  export GITMI_ATTN_PW=2; run pw2_$i
done
export BENCH_GEMM_IMPL=32777
for i in 1 2; do
  unset GITMI_ATTN_PW; run all256_pw1_$i
  export GITMI_ATTN_PW=4; run all256_pw4_$i
done
t done
