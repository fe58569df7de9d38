#!/bin/bash
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "dgemm or vocab or attention_decode" > gpurun_out/d_ops.txt 2>&1; echo "ops rc=$?"; tail -n 3 gpurun_out/d_ops.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -s -k "tiny_f32 or tiny_bf16 or ragged or answer or full_batch or full_size" > gpurun_out/d_par.txt 2>&1; echo "parity rc=$?"; tail -n 5 gpurun_out/d_par.txt
timeout 300 python tools/dgemm_bench.py 64 > gpurun_out/d_dgemm_bench.txt 2>&1; echo "rc=$?"; grep -v amdgpu gpurun_out/d_dgemm_bench.txt
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/d_bench.txt 2>&1; echo "bench rc=$?"; tail -n 1 gpurun_out/d_bench.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline_decode']['avg_step_ms'], d['roofline_decode']['frac'], d['phases_ms'], d['parity'])"
