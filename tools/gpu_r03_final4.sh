#!/bin/bash
# Round 3, the row-walking wide chain GEMM (beam batches / decode groups) and the member-order guard: unit + group tests, A/B
# of the kernel in the beam-4 and decode-group schedules, then the closing sequence for the new csrc -- PMC passes (copied
# to profiles/ first), full GPU suite, smoke, default line, bench lines of the other configs
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-r03_zzz}; T0=$(date +%s); t() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
t "unit + group tests"; timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_group.py -q --tb=short -p no:cacheprovider -x -k "dgemm or group" 2>&1 | tail -n 4
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('parity') or {}; print(d['value'], 'captions/s', d['ms_per_step'], 'ms/pass  latency', d['batch_latency_ms']['median'], 'dec step', d['roofline_decode']['avg_step_ms'], d['roofline_decode'].get('solo_policy', {}).get('avg_step_ms'), 'ids==solo', d.get('timed_ids_equal_solo'), 'identical', p.get('identical'), p.get('ok'))"; }
run() { local n=$1 nw=$2; shift; shift; local f=gpurun_out/${TAG}_ab_$n.json; GITMI_DGEMM_NO_ROW_WALK=$nw timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 8 "$@" 2>gpurun_out/${TAG}_err_$n.txt | tail -n 1 > $f; t "$n: $(line < $f 2>&1 | tail -n 1)"; [ -s $f ] || tail -n 4 gpurun_out/${TAG}_err_$n.txt; }
for i in 1 2; do
  run beam_blocks_$i 1 --search beam
  run beam_walk_$i 0 --search beam
done
run group2_blocks 1 --decode-group 2
run group2_walk 0 --decode-group 2
run group4_walk 0 --decode-group 4 --contexts 8
t "pmc"; bash tools/gpu_pmc.sh > gpurun_out/${TAG}_pmc.log 2>&1; cp gpurun_out/pmc_summary.tsv gpurun_out/${TAG}_pmc_summary.tsv; cp gpurun_out/pmc_summary.tsv profiles/${TAG}_pmc_summary.tsv; rm -rf gpurun_out/pmc; grep csrc_sha gpurun_out/${TAG}_pmc_summary.tsv
rm -f gpurun_out/parity_measured.jsonl
t "pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/${TAG}_pytest_gpu.txt 2>&1; echo "rc=$?"; tail -n 3 gpurun_out/${TAG}_pytest_gpu.txt | cut -c1-250
cp gpurun_out/parity_measured.jsonl gpurun_out/${TAG}_parity_measured.jsonl 2>/dev/null
t "smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.txt 2>&1; tail -n 1 gpurun_out/${TAG}_smoke.txt
t "default bench"; timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "rc=$?"; python -c "import json; d=json.loads(open('gpurun_out/${TAG}_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac'], d['roofline']['traffic_source'], d['roofline']['traffic_stale'], d['parity']['identical'])"
cfg() { local name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --steps 20 --warmup 4 "$@" 2> gpurun_out/${TAG}_${name}.err | tail -n 1 > gpurun_out/${TAG}_${name}_bench.json; python -c "import json; d=json.load(open('gpurun_out/${TAG}_${name}_bench.json')); p=d.get('parity') or {}; print('$name', d['dtype'], d['value'], d['ms_per_step'], 'ms | gemm', d['roofline']['frac'], 'decode frac', d['roofline_decode']['frac'], 'step', d['roofline_decode']['avg_step_ms'], '| parity', p.get('identical'), '/', p.get('rows'), p.get('ok'))"; }
cfg beam4_bf16 --search beam
cfg beam4_f16 --search beam --precision f16
t done
