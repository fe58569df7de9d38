#!/bin/bash
# unprofiled bench lines of the other BASELINE configs: bash tools/gpu_configs.sh [tag]
TAG=${1:-cfg}
mkdir -p gpurun_out
run() {
  local name=$1; shift
  timeout 400 python bench.py --no-cpu-baseline --steps 20 --warmup 4 "$@" 2> gpurun_out/${TAG}_${name}.err | tail -n 1 > gpurun_out/${TAG}_${name}_bench.json
  python -c "import sys,json; d=json.load(open('gpurun_out/${TAG}_${name}_bench.json')); print('$name', d['value'], d['unit'], d['ms_per_step'], 'ms/step', d['config']['workload'], 'gemm', d['roofline']['frac'], 'decode step ms', d.get('roofline_decode',{}).get('avg_step_ms'), 'parity', d.get('parity'))"
}
run beam4 --search beam
run large_b32 --model GIT_LARGE_COCO --batch 32
run vatex_b16 --model GIT_BASE_VATEX --frames 6 --batch 16
