#!/bin/bash
# Round 3: decode groups (gitmi_set_decode_group / gitmi_group_decode) -- parity tests, then the mixed bench with groups of
# 2 / 4 requests per decode chain against the default schedule (interleaved)
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-r03_n}; T0=$(date +%s); t() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
t "group tests"; timeout 600 python -m pytest tests/test_gpu_group.py -q --tb=short -p no:cacheprovider -x > gpurun_out/${TAG}_group_tests.txt 2>&1; tail -n 15 gpurun_out/${TAG}_group_tests.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('parity') or {}; r=d['roofline_decode']; print(d['value'], 'captions/s', d['ms_per_step'], 'ms/pass  latency', d['batch_latency_ms'], 'dec step', r['avg_step_ms'], 'frac', r['frac'], 'ids==solo', d.get('timed_ids_equal_solo'), 'identical', p.get('identical'))"; }
run() { # name args...
  local n=$1; shift
  local f=gpurun_out/${TAG}_bench_$n.json
  timeout 300 python bench.py --no-cpu-baseline --steps 48 --warmup 8 "$@" 2>gpurun_out/${TAG}_err_$n.txt | tail -n 1 > $f
  t "$n: $(line < $f 2>&1 | tail -n 1)"
  [ -s $f ] || tail -n 5 gpurun_out/${TAG}_err_$n.txt
}
run default_1
run g2_c4_e2 --decode-group 2 --contexts 4
run g2_c6_e2 --decode-group 2 --contexts 6
run g2_c6_e3 --decode-group 2 --contexts 6 --encoder-chains 3
run g2_c4_e1 --decode-group 2 --contexts 4 --encoder-chains 1
run g4_c4_e2 --decode-group 4 --contexts 4
run g4_c8_e2 --decode-group 4 --contexts 8
run g3_c6_e2 --decode-group 3 --contexts 6
run default_2
export BENCH_GEMM_IMPL=32777
run g2_c4_e2_all256 --decode-group 2 --contexts 4
run g4_c8_e2_all256 --decode-group 4 --contexts 8
t done
