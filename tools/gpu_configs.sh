#!/bin/bash
# The other BASELINE.json configs on one GPU (parity-test cases; recorded for reference, not the headline line).
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "task_function" 2>&1 | tail -8
run() { echo "== $*"; timeout 400 python bench.py --no-cpu-baseline --steps 12 --warmup 3 "$@" 2>&1 | grep -v amdgpu | tail -1 | tee -a gpurun_out/configs.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], 'ms/step', d['config']['workload'], d['roofline']['achieved'], 'TF', d['phases_ms'])"; }
rm -f gpurun_out/configs.jsonl
run --search beam
run --model GIT_LARGE_COCO --batch 32
run --model GIT_BASE_VATEX --frames 6 --batch 16
run --batch 256 --contexts 2
