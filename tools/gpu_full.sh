#!/bin/bash
# Full GPU parity suite + bench + rocprof kernel stats.
set -u
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.txt 2>&1; echo "rc=$?"; tail -30 gpurun_out/pytest_gpu.txt
echo "== bench"; timeout 900 python bench.py > gpurun_out/bench.txt 2>&1; echo "rc=$?"; tail -2 gpurun_out/bench.txt
echo "== rocprof"; cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.txt 2>&1; echo "rc=$?"; cd $GRAFT_REPO_ROOT; ls -R gpurun_out/prof | head -20
