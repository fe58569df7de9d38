#!/bin/bash
# Round 3: serving policy (gitmi_set_shared_device: 256-row GEMM tiles everywhere + two pairs per attention workgroup) and
# sc1 output stores -- full GPU suite, then interleaved A/B of the policy in the mixed bench, then a small sweep under it
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-r03_s}; T0=$(date +%s); t() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
rm -f gpurun_out/parity_measured.jsonl
t "full GPU suite"; timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/${TAG}_pytest_gpu.txt 2>&1; tail -n 6 gpurun_out/${TAG}_pytest_gpu.txt
cp gpurun_out/parity_measured.jsonl gpurun_out/${TAG}_parity_measured.jsonl 2>/dev/null
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('parity') or {}; r=d['roofline']; print(d['value'], 'captions/s', d['ms_per_step'], 'ms/pass  latency', d['batch_latency_ms']['median'], 'gemm us', round(r['avg_launch_ms']*1e3,1), 'frac', r['frac'], 'dec step', d['roofline_decode']['avg_step_ms'], 'ids==solo', d.get('timed_ids_equal_solo'), 'identical', p.get('identical'))"; }
run() { # name args...
  local n=$1; shift
  local f=gpurun_out/${TAG}_bench_$n.json
  timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 8 "$@" 2>gpurun_out/${TAG}_err_$n.txt | tail -n 1 > $f
  t "$n: $(line < $f 2>&1 | tail -n 1)"
  [ -s $f ] || tail -n 5 gpurun_out/${TAG}_err_$n.txt
}
for i in 1 2 3; do
  run solo_policy_$i --solo-policy
  run shared_policy_$i
done
run shared_c3_e2 --contexts 3
run shared_c5_e2 --contexts 5
run shared_c6_e2 --contexts 6
run shared_c6_e3 --contexts 6 --encoder-chains 3
run shared_c4_e3 --contexts 4 --encoder-chains 3
run shared_g2 --decode-group 2
run shared_beam --search beam
run solo_beam --search beam --solo-policy
t done
