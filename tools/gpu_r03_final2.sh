#!/bin/bash
# Round 3, after the last kernel change (attention kernel by geometry, packing capped by the pair count): the full GPU suite,
# smoke, the default line (with the CPU baseline) and the lines of the other BASELINE configs in both operand builds.
# Kernel statistics / PMC of tools/gpu_r03_final.sh (r03_z) stay valid: the default configuration runs the same kernels.
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-r03_zz}; T0=$(date +%s); t() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
rm -f gpurun_out/parity_measured.jsonl
t "pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/${TAG}_pytest_gpu.txt 2>&1; echo "rc=$?"; tail -n 3 gpurun_out/${TAG}_pytest_gpu.txt | cut -c1-250
cp gpurun_out/parity_measured.jsonl gpurun_out/${TAG}_parity_measured.jsonl 2>/dev/null
t "smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.txt 2>&1; tail -n 2 gpurun_out/${TAG}_smoke.txt
t "default bench"; timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "rc=$?"; tail -n 1 gpurun_out/${TAG}_bench.json | cut -c1-300
run() { local name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --steps 20 --warmup 4 "$@" 2> gpurun_out/${TAG}_${name}.err | tail -n 1 > gpurun_out/${TAG}_${name}_bench.json; python -c "import json; d=json.load(open('gpurun_out/${TAG}_${name}_bench.json')); p=d.get('parity') or {}; print('$name', d['dtype'], d['value'], d['ms_per_step'], 'ms | gemm', d['roofline']['frac'], 'decode frac', d['roofline_decode']['frac'], 'step', d['roofline_decode']['avg_step_ms'], '| parity', p.get('identical'), '/', p.get('rows'), p.get('ok'))"; }
for pr in bf16 f16; do
  run base_${pr} --precision $pr
  run beam4_${pr} --search beam --precision $pr
  run large_b32_${pr} --model GIT_LARGE_COCO --batch 32 --precision $pr
  run vatex_b16_${pr} --model GIT_BASE_VATEX --frames 6 --batch 16 --precision $pr
done
run base_200steps --steps 200 --warmup 8
t done
