#!/bin/bash
# Round 4, call D: decode groups again, now that the chain kernels walk (fixed workgroup counts whatever the row count)
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 HSA_ENABLE_IPC_MODE_LEGACY=0
AB_TIMEOUT=90 AB_STEPS=48 AB_WARMUP=8 bash tools/gpu_ab.sh r04_d 2 \
  "base:" \
  "dg2: -- --decode-group 2" \
  "dg2c8: -- --decode-group 2 --contexts 8" \
  "dg4c8: -- --decode-group 4 --contexts 8" \
  "dg2c6: -- --decode-group 2 --contexts 6 --encoder-chains 2"
