#!/bin/bash
set -u; export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_preprocess.py -m gpu -q -x -k "varres or vqa or minmax" -p no:cacheprovider > gpurun_out/varres.txt 2>&1
echo rc=$?
head -60 gpurun_out/varres.txt
