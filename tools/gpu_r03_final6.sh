#!/bin/bash
# Round 3: wide chain GEMMs with two strips per workgroup, one after the other (dgemm_wide2_kernel) -- unit + policy tests,
# interleaved A/B in the mixed schedule
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-r03_w2}; T0=$(date +%s); t() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
t "tests"; timeout 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_group.py -q --tb=short -p no:cacheprovider -x -k "dgemm_qkv or policy" 2>&1 | tail -n 3
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('parity') or {}; print(d['value'], 'captions/s', d['ms_per_step'], 'ms/pass  latency', d['batch_latency_ms']['median'], 'dec step', d['roofline_decode']['avg_step_ms'], 'ids==solo', d.get('timed_ids_equal_solo'), 'identical', p.get('identical'), p.get('ok'))"; }
run() { local n=$1 v=$2; shift; shift; local f=gpurun_out/${TAG}_ab_$n.json; GITMI_DGEMM_WIDE2=$v timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 8 "$@" 2>gpurun_out/${TAG}_err_$n.txt | tail -n 1 > $f; t "$n: $(line < $f 2>&1 | tail -n 1)"; [ -s $f ] || tail -n 4 gpurun_out/${TAG}_err_$n.txt; }
for i in 1 2 3; do
  run one_strip_$i 0
  run two_strips_$i 1
done
t done
