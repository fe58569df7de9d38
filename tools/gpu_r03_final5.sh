#!/bin/bash
# Round 3, closing call for the final csrc: PMC passes (copied to profiles/ first), the unit tests of the chain GEMMs, default
# line (CPU baseline included), beam-4 line with and without the serving policy
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-r03_final}; T0=$(date +%s); t() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
t "pmc"; bash tools/gpu_pmc.sh > gpurun_out/${TAG}_pmc.log 2>&1; cp gpurun_out/pmc_summary.tsv gpurun_out/${TAG}_pmc_summary.tsv; cp gpurun_out/pmc_summary.tsv profiles/${TAG}_pmc_summary.tsv; rm -rf gpurun_out/pmc; grep csrc_sha gpurun_out/${TAG}_pmc_summary.tsv
t "chain GEMM + group + policy tests"; timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_group.py -q --tb=short -p no:cacheprovider -k "dgemm or group or policy" 2>&1 | tail -n 2
t "default bench"; timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "rc=$?"; python -c "import json; d=json.loads(open('gpurun_out/${TAG}_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['solo_policy'], d['roofline']['traffic_source'], d['roofline']['traffic_stale'], d['roofline_decode']['frac'], d['roofline_decode']['solo_policy'], d['parity']['identical'], d['cpu_baseline']['value'])"
cfg() { local name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --steps 20 --warmup 4 "$@" 2> gpurun_out/${TAG}_${name}.err | tail -n 1 > gpurun_out/${TAG}_${name}_bench.json; python -c "import json; d=json.load(open('gpurun_out/${TAG}_${name}_bench.json')); p=d.get('parity') or {}; print('$name', d['dtype'], d['value'], d['ms_per_step'], 'ms | gemm', d['roofline']['frac'], 'decode step', d['roofline_decode']['avg_step_ms'], (d['roofline_decode'].get('solo_policy') or {}).get('avg_step_ms'), '| parity', p.get('identical'), '/', p.get('rows'), p.get('ok'))"; }
cfg beam4_bf16 --search beam
cfg beam4_solo_policy --search beam --solo-policy
cfg base_200steps --steps 200 --warmup 8
t done
