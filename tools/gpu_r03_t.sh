#!/bin/bash
# Round 3: N = 768 chain GEMMs with 32 / 64 rows per workgroup (a half / a quarter of the workgroups) in the mixed schedule
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-r03_t}; T0=$(date +%s); t() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
for rows in 64 32; do
  t "parity slice with GITMI_DGEMM_ROWS=$rows"
  GITMI_DGEMM_ROWS=$rows timeout 900 python -m pytest tests/test_gpu_parity.py -q --tb=short -p no:cacheprovider -x -k "bf16 or f16 or full" 2>&1 | tail -n 3
done
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('parity') or {}; r=d['roofline']; print(d['value'], 'captions/s', d['ms_per_step'], 'ms/pass  latency', d['batch_latency_ms']['median'], 'dec step', d['roofline_decode']['avg_step_ms'], 'ids==solo', d.get('timed_ids_equal_solo'), 'identical', p.get('identical'))"; }
run() { # name rows args...
  local n=$1 rows=$2; shift; shift
  local f=gpurun_out/${TAG}_bench_$n.json
  GITMI_DGEMM_ROWS=$rows timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 8 "$@" 2>gpurun_out/${TAG}_err_$n.txt | tail -n 1 > $f
  t "$n: $(line < $f 2>&1 | tail -n 1)"
  [ -s $f ] || tail -n 5 gpurun_out/${TAG}_err_$n.txt
}
for i in 1 2 3; do
  run rows16_$i 16
  run rows64_$i 64
  run rows32_$i 32
done
run beam_rows16 16 --search beam
run beam_rows64 64 --search beam
run beam_rows32 32 --search beam
t done
