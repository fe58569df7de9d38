#!/bin/bash
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
run() { timeout 300 python bench.py --no-cpu-baseline --contexts $1 --steps 48 --warmup 8 2>&1 | grep -v amdgpu | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for c in 3 4 5 6 8; do echo "== contexts $c"; run $c; done
