#!/bin/bash
# Does one engine pass over 128 rows (two 64-image requests decoded together) beat two 64-row passes?
# bash tools/gpu_coalesce.sh  -> captions/s, ms per pass, decode step of B=64 (default) vs B=128 at 2/3/4 contexts
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
run() {
  timeout 120 python bench.py --no-cpu-baseline --steps "$1" --warmup 4 "${@:2}" 2>/dev/null | tail -n 1 | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], 'captions/s', d['ms_per_step'], 'ms/pass', d['config']['workload'], 'ctx', d['config']['contexts_in_flight'], 'chains', d['config']['encoder_chains'], 'decode step ms', d['roofline_decode']['avg_step_ms'], 'enc+prefill', d['phases_ms']['graph_encode_prefill_ms'], 'lat', d['batch_latency_ms']['median'])"
}
echo "== B=64 default"; run 40
echo "== B=128 ctx2 chains1"; run 20 --batch 128 --contexts 2 --encoder-chains 1
echo "== B=128 ctx2 free"; run 20 --batch 128 --contexts 2 --free-run
echo "== B=128 ctx3 chains1"; run 20 --batch 128 --contexts 3 --encoder-chains 1
echo "== B=128 ctx4 chains2"; run 20 --batch 128 --contexts 4 --encoder-chains 2
echo "== B=64 default again"; run 40
