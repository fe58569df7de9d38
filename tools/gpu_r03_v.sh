#!/bin/bash
# Round 3: decode attention with ONE wave per (sentence, head) pair (half the resident waves, two memory round trips)
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-r03_v}; T0=$(date +%s); t() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
t "parity (bf16 full-batch cases) with GITMI_ATTN_NH=1"
GITMI_ATTN_NH=1 timeout 600 python -m pytest tests/test_gpu_parity.py -q --tb=short -p no:cacheprovider -x -k "test_full_batch_ids_against_reference and (bench_b64_greedy or base_b64_beam4)" 2>&1 | tail -n 3
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('parity') or {}; print(d['value'], 'captions/s', d['ms_per_step'], 'ms/pass  latency', d['batch_latency_ms']['median'], 'dec step', d['roofline_decode']['avg_step_ms'], 'ids==solo', d.get('timed_ids_equal_solo'), 'identical', p.get('identical'))"; }
run() { # name nh args...
  local n=$1 nh=$2; shift; shift
  local f=gpurun_out/${TAG}_bench_$n.json
  GITMI_ATTN_NH=$nh timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 8 "$@" 2>gpurun_out/${TAG}_err_$n.txt | tail -n 1 > $f
  t "$n: $(line < $f 2>&1 | tail -n 1)"
  [ -s $f ] || tail -n 5 gpurun_out/${TAG}_err_$n.txt
}
for i in 1 2 3; do
  run nh2_$i 2
  run nh1_$i 1
done
run beam_nh2 2 --search beam
run beam_nh1 1 --search beam
t done
