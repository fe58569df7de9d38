#!/bin/bash
# Round 3, sixth GPU call: wide LayerNorm kernel (unit + engine parity), schedule sweep of the mixed bench
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD; TAG=${1:-r03_f}; T0=$(date +%s); t() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
t "unit + parity tests"; timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity.py tests/test_gpu_stream16.py -q --tb=short -p no:cacheprovider -k "layernorm or bf16 or full_batch or scripted or beam" > gpurun_out/${TAG}_pytest.txt 2>&1; echo "rc=$?"; tail -n 8 gpurun_out/${TAG}_pytest.txt | cut -c1-250
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('parity') or {}; print(d['value'], 'captions/s', d['ms_per_step'], 'ms/pass | gemm frac', d['roofline']['frac'], '| enc+prefill', d['phases_ms']['graph_encode_prefill_ms'], 'decode', d['phases_ms']['graph_decode_ms'], 'lat', d['batch_latency_ms']['median'], '| parity', p.get('identical'), p.get('ok'))"; }
for i in 1 2; do
  for cfg in "4 2" "5 2" "6 2" "6 3" "3 2" "8 2" "8 3"; do
    set -- $cfg
    t "bench contexts=$1 chains=$2 ($i)"; timeout 600 python bench.py --no-cpu-baseline --steps 40 --warmup 8 --contexts $1 --encoder-chains $2 2> gpurun_out/${TAG}_sweep.err | tee gpurun_out/${TAG}_bench_c$1_e$2_$i.json | line
  done
done
t "beam"; timeout 600 python bench.py --no-cpu-baseline --search beam --steps 12 --warmup 3 2> gpurun_out/${TAG}_beam.err | tee gpurun_out/${TAG}_bench_beam.json | line
t done
