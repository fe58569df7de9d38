#!/bin/bash
# Round 4, call F: the streaming decode attention (K/V through an LDS ring) -- unit test (bitwise == the one-wave register
# kernel), then interleaved A/B of the whole bench over its workgroup count, solo decode-step figures
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-r04_f}; T0=$(date +%s); t() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
t "op test"; timeout 300 python -m pytest tests/test_gpu_ops.py -q --tb=short -p no:cacheprovider -x -k "attention_decode" 2>&1 | tail -n 8 | cut -c1-300
AB_TIMEOUT=90 AB_STEPS=40 AB_WARMUP=8 bash tools/gpu_ab.sh $TAG 2 \
  "base:GITMI_ATTN_STREAM=0" \
  "st48:GITMI_ATTN_STREAM=48" \
  "st64:GITMI_ATTN_STREAM=64" \
  "st96:GITMI_ATTN_STREAM=96" \
  "st128:GITMI_ATTN_STREAM=128" \
  "st192:GITMI_ATTN_STREAM=192"
t "solo decode step"
for v in 0 48 64 96 192; do
  GITMI_ATTN_STREAM=$v timeout 200 python bench.py --experiment --no-cpu-baseline --contexts 1 --steps 10 --warmup 2 2>/dev/null | tail -n 1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('stream wgs $v', 'solo ms/pass', d['ms_per_step'], 'decode step', d['roofline_decode'].get('avg_step_ms'), 'identical', (d.get('parity') or {}).get('identical'), 'wide', ((d.get('parity') or {}).get('wide_margin') or {}).get('identical'))"
done
t done
t "new parity tests"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream16.py tests/test_preprocess.py -q --tb=short -p no:cacheprovider -k "keep_best or wide_margin or serving_policy or scripted or stream or preproc or fp16_stream or full_batch" 2>&1 | tail -n 12 | cut -c1-300
cp gpurun_out/parity_measured.jsonl gpurun_out/${TAG}_parity_measured.jsonl 2>/dev/null
t done
