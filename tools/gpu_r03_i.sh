#!/bin/bash
# Round 3, ninth GPU call: the fp16-operand build (libgitmi_f16.so) -- parity tests, bench line, smoke
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD; TAG=${1:-r03_i}; T0=$(date +%s); t() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
rm -f gpurun_out/parity_measured.jsonl
t "smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 4
t "f16 + full-batch parity tests"; timeout 1200 python -m pytest tests/test_gpu_parity.py -q --tb=short -p no:cacheprovider -k "f16_operand or full_batch" > gpurun_out/${TAG}_pytest.txt 2>&1; echo "rc=$?"; tail -n 25 gpurun_out/${TAG}_pytest.txt | cut -c1-250
cp gpurun_out/parity_measured.jsonl gpurun_out/${TAG}_parity_measured.jsonl 2>/dev/null
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('parity') or {}; print(d['dtype'], d['value'], 'captions/s', d['ms_per_step'], 'ms/pass | gemm frac', d['roofline']['frac'], '| enc+prefill', d['phases_ms']['graph_encode_prefill_ms'], 'decode', d['phases_ms']['graph_decode_ms'], '| parity', p.get('identical'), p.get('ok'), p.get('logit_err'), p.get('violation'))"; }
for i in 1 2; do
  for pr in bf16 f16; do
    t "bench $pr ($i)"; timeout 600 python bench.py --no-cpu-baseline --steps 40 --warmup 8 --precision $pr 2> gpurun_out/${TAG}_$pr.err | tee gpurun_out/${TAG}_bench_${pr}_$i.json | line
  done
done
t "bench f16 beam"; timeout 600 python bench.py --no-cpu-baseline --search beam --steps 12 --warmup 3 --precision f16 2> gpurun_out/${TAG}_beam.err | tee gpurun_out/${TAG}_bench_f16_beam.json | line
t "bench f16 large"; timeout 600 python bench.py --no-cpu-baseline --model GIT_LARGE_COCO --batch 32 --steps 12 --warmup 3 --precision f16 2> gpurun_out/${TAG}_large.err | tee gpurun_out/${TAG}_bench_f16_large.json | line
t "bench f16 vatex"; timeout 600 python bench.py --no-cpu-baseline --model GIT_BASE_VATEX --frames 6 --batch 16 --steps 12 --warmup 3 --precision f16 2> gpurun_out/${TAG}_vatex.err | tee gpurun_out/${TAG}_bench_f16_vatex.json | line
t done
