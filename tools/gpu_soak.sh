#!/bin/bash
# race screen: repeat the GEMM A/B (bitwise equality between variants and between reruns) and the op tests
set -u; export PYTHONUNBUFFERED=1
for i in 1 2 3; do timeout 300 python tools/gemm_bench.py 2,9 2>&1 | grep -v amdgpu | awk '{ok=1; for(i=1;i<=NF;i++){ if($i ~ /^dbase=/ && $i != "dbase=0") ok=0; if($i ~ /^drerun=/ && $i != "drerun=0") ok=0}; print (ok?"OK  ":"BAD ") $1, $0 ~ /impl9/ ? "" : ""}' ; done
for i in 1 2; do timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider -k "p8 or attention" 2>&1 | tail -1; done
