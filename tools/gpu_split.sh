#!/bin/bash
set -u; export PYTHONUNBUFFERED=1
run() { timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 8 "$@" 2>&1 | grep -v amdgpu | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
echo "== full T=20 ctx4"; run
echo "== T=2 (encoder+prefill+1 step) ctx4"; run --max-steps 2
echo "== T=2 ctx1"; run --max-steps 2 --contexts 1
echo "== full ctx1"; run --contexts 1
