#!/bin/bash
set -u; export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "gemm" -p no:cacheprovider 2>&1 | tail -5
timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep -v amdgpu | tail -1 | tee gpurun_out/bench_p8.txt
