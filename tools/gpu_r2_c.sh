#!/bin/bash
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 300 python tools/dgemm_bench.py 64 > gpurun_out/c_dgemm_bench.txt 2>&1; echo "rc=$?"; cat gpurun_out/c_dgemm_bench.txt | grep -v amdgpu
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/c_bench.txt 2>&1; echo "bench rc=$?"; tail -n 1 gpurun_out/c_bench.txt | cut -c1-1500
