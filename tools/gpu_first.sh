#!/bin/bash
# First-contact GPU run: every stage under its own timeout, all output under gpurun_out/.
set -u
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9|Compute Unit" | head -6 > gpurun_out/rocminfo.txt
nproc > gpurun_out/nproc.txt; lscpu | grep -E "Model name|^CPU\(s\)|Socket|Thread" >> gpurun_out/nproc.txt
echo "== ops"; timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_ops.txt 2>&1; echo "ops rc=$?"; tail -15 gpurun_out/pytest_ops.txt
echo "== parity tiny"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "tiny or scripted or step_logits" > gpurun_out/pytest_tiny.txt 2>&1; echo "tiny rc=$?"; tail -25 gpurun_out/pytest_tiny.txt
echo "== smoke"; timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.txt 2>&1; echo "smoke rc=$?"; tail -5 gpurun_out/smoke.txt
echo "== bench eager"; timeout 600 python bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline > gpurun_out/bench_eager.txt 2>&1; echo "rc=$?"; tail -3 gpurun_out/bench_eager.txt
echo "== bench graph"; timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_graph.txt 2>&1; echo "rc=$?"; tail -3 gpurun_out/bench_graph.txt
