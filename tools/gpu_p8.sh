#!/bin/bash
set -u; export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
timeout 600 python tools/gemm_bench.py ${1:-2,6,9} 2>&1 | grep -v amdgpu | tee gpurun_out/gemm_p8.txt
