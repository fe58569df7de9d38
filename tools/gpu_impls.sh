#!/bin/bash
set -u; export PYTHONUNBUFFERED=1
run() { timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 8 2>&1 | grep -v amdgpu | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['phases_ms'])"; }
for impl in 9 2 0; do echo "== GITMI_GEMM_IMPL=$impl"; GITMI_GEMM_IMPL=$impl run; done
