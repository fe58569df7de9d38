#!/bin/bash
# round 2, first GPU pass: new decode-chain kernels, search step kernel, tiny parity, ragged prefixes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "dgemm or vocab or attention_decode" > gpurun_out/a_ops.txt 2>&1; echo "ops rc=$?" >> gpurun_out/a_ops.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "scripted" > gpurun_out/a_scripted.txt 2>&1; echo "rc=$?" >> gpurun_out/a_scripted.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "tiny_f32 or tiny_bf16" -s > gpurun_out/a_tiny.txt 2>&1; echo "rc=$?" >> gpurun_out/a_tiny.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "ragged or answer or bare_tensor or rccl" -s > gpurun_out/a_new.txt 2>&1; echo "rc=$?" >> gpurun_out/a_new.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/a_bench.txt 2>&1; echo "rc=$?" >> gpurun_out/a_bench.txt
tail -5 gpurun_out/a_ops.txt gpurun_out/a_scripted.txt gpurun_out/a_tiny.txt gpurun_out/a_new.txt gpurun_out/a_bench.txt
