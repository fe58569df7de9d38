#!/bin/bash
# Round 4, call A: the walking vocabulary head, N-strip chain GEMMs, looping decode attention -- op / parity tests, then
# interleaved A/B of the whole bench on one box
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-r04_a}; T0=$(date +%s); t() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
t "op tests"; timeout 600 python -m pytest tests/test_gpu_ops.py -q --tb=short -p no:cacheprovider -x -k "vocab or dgemm or attention_decode" 2>&1 | tail -n 4 | cut -c1-300
t "parity slice"; timeout 600 python -m pytest tests/test_gpu_parity.py -q --tb=short -p no:cacheprovider -x -k "wide_margin or full_batch_ids or base_greedy or tiny_beam4 or policy" 2>&1 | tail -n 6 | cut -c1-300
cp gpurun_out/parity_measured.jsonl gpurun_out/${TAG}_parity_measured.jsonl 2>/dev/null
B="GITMI_VOCAB_WGS=0 GITMI_DGEMM_STRIPS=2"
AB_STEPS=40 AB_WARMUP=8 bash tools/gpu_ab.sh $TAG 2 \
  "base:$B" \
  "v60:GITMI_VOCAB_WGS=60 GITMI_DGEMM_STRIPS=2" \
  "v40:GITMI_VOCAB_WGS=40 GITMI_DGEMM_STRIPS=2" \
  "s4:GITMI_VOCAB_WGS=0 GITMI_DGEMM_STRIPS=4" \
  "ppw2:$B GITMI_ATTN_PPW=2" \
  "combo:GITMI_VOCAB_WGS=60 GITMI_DGEMM_STRIPS=4 GITMI_ATTN_PPW=2"
AB_STEPS=40 AB_WARMUP=8 bash tools/gpu_ab.sh ${TAG}_beam 1 \
  "base:$B -- --search beam" \
  "v60:GITMI_VOCAB_WGS=60 GITMI_DGEMM_STRIPS=2 -- --search beam"
t "solo decode step per variant (one context: roofline_decode)"
for v in "GITMI_VOCAB_WGS=0" "GITMI_VOCAB_WGS=60" "GITMI_DGEMM_STRIPS=4" "GITMI_ATTN_PPW=2 GITMI_ATTN_PW=8"; do
  env $v timeout 200 python bench.py --experiment --no-cpu-baseline --contexts 1 --steps 10 --warmup 2 2>/dev/null | tail -n 1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('$v', 'solo ms/pass', d['ms_per_step'], 'decode step', d['roofline_decode'].get('avg_step_ms'), 'identical', (d.get('parity') or {}).get('identical'))"
done
t done
