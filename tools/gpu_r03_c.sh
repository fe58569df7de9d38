#!/bin/bash
# Round 3, third GPU call: diagnosis of the persistent GEMM + the trie tests + the GIT_LARGE attention fix
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-r03_c}; T0=$(date +%s); t() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
t "p9 diagnosis"; timeout 600 python tools/gemm_p9_diag.py 2>&1 | grep -v amdgpu | tee gpurun_out/${TAG}_gemm_p9_diag.txt
t "trie + attention + search tests"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ops.py -q --tb=short -p no:cacheprovider -k "trie or attention or attn or scripted or full_batch" > gpurun_out/${TAG}_pytest.txt 2>&1; echo "rc=$?"; tail -n 15 gpurun_out/${TAG}_pytest.txt | cut -c1-250
t "large bench"; timeout 600 python bench.py --no-cpu-baseline --model GIT_LARGE_COCO --batch 32 --steps 12 --warmup 3 2> gpurun_out/${TAG}_large.err | tail -n 1 > gpurun_out/${TAG}_bench_large.json; python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench_large.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['parity'])"
t done
