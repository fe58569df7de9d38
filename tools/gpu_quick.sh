#!/bin/bash
set -u; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "task_function or video or vatex or tiny_greedy" 2>&1 | tail -5
run() { echo "== $*"; timeout 400 python bench.py --no-cpu-baseline --steps 12 --warmup 3 "$@" 2>&1 | grep -v amdgpu | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], 'ms/step', d['config']['workload'], d['roofline']['achieved'], 'TF', d['phases_ms'])"; }
run --model GIT_BASE_VATEX --frames 6 --batch 16
run
