#!/bin/bash
# PMC passes for the current csrc (-> profiles/<tag>_pmc_summary.tsv), then the mixed schedule vs the phased one
# (bench.py --phased G: encoders of a group of G batches first, then its decode chains side by side)
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-phased}; T0=$(date +%s)
stamp() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
stamp pmc; bash tools/gpu_pmc.sh > gpurun_out/${TAG}_pmc.log 2>&1
cp gpurun_out/pmc_summary.tsv gpurun_out/${TAG}_pmc_summary.tsv && cp gpurun_out/pmc_summary.tsv profiles/${TAG}_pmc_summary.tsv
rm -rf gpurun_out/pmc; sed -n 3p gpurun_out/${TAG}_pmc_summary.tsv
run() {
  stamp "bench $*"
  timeout 60 python bench.py --no-cpu-baseline --steps 40 --warmup 8 "$@" 2> gpurun_out/${TAG}_err.txt | tail -n 1 > gpurun_out/${TAG}_last.json
  python -c "
import sys,json
d=json.loads(open('gpurun_out/${TAG}_last.json').read().strip().splitlines()[-1])
print(d['value'], 'captions/s', d['ms_per_step'], 'ms/pass |', d['config']['schedule'][:24], '| ctx', d['config']['contexts_in_flight'], 'lat', d['batch_latency_ms'], 'ids==solo', d.get('timed_ids_equal_solo'), 'parity', (d.get('parity') or {}).get('identical'), (d.get('parity') or {}).get('ok'), 'stale', d['roofline'].get('traffic_stale'))" || tail -n 5 gpurun_out/${TAG}_err.txt
  cat gpurun_out/${TAG}_last.json >> gpurun_out/${TAG}_bench_lines.jsonl
}
run --phased 4
run
run --phased 8
run --phased 4 --encoder-chains 4
stamp done
