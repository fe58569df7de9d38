#!/bin/bash
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 300 python tools/dgemm_bench.py 64 2>&1 | grep -E "attn_decode" 
