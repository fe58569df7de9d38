#!/bin/bash
# Round 3, closing call for the FINAL csrc (two-strip wide chain GEMM under the serving policy): PMC passes (copied to
# profiles/ first), default line, full GPU suite
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-r03_last}; T0=$(date +%s); t() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
t "pmc"; bash tools/gpu_pmc.sh > gpurun_out/${TAG}_pmc.log 2>&1; cp gpurun_out/pmc_summary.tsv gpurun_out/${TAG}_pmc_summary.tsv; cp gpurun_out/pmc_summary.tsv profiles/${TAG}_pmc_summary.tsv; rm -rf gpurun_out/pmc; grep csrc_sha gpurun_out/${TAG}_pmc_summary.tsv
t "default bench"; timeout 300 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "rc=$?"; python -c "import json; d=json.loads(open('gpurun_out/${TAG}_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['solo_policy'], d['roofline']['traffic_source'], d['roofline']['traffic_stale'], d['roofline_decode']['frac'], d['roofline_decode']['solo_policy'], d['parity']['identical'], d['cpu_baseline']['value'])"
rm -f gpurun_out/parity_measured.jsonl
t "pytest -m gpu"; timeout 400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/${TAG}_pytest_gpu.txt 2>&1; echo "rc=$?"; tail -n 3 gpurun_out/${TAG}_pytest_gpu.txt | cut -c1-250
cp gpurun_out/parity_measured.jsonl gpurun_out/${TAG}_parity_measured.jsonl 2>/dev/null
t done
