#!/bin/bash
# A/B of the fp32 epilogue of the p8 GEMM: impl 65545 (= 9 | 256 << 8) LDS-staged, 9 direct from the accumulators
# (isolated launches with bitwise comparison against the staged variant, GEMM unit tests, then the whole bench in situ)
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
GEMM_BENCH_SHAPES=vit.out,vit.c_proj,patch timeout 200 python tools/gemm_bench.py 65545,9 2>&1 | grep -v amdgpu
timeout 200 python -m pytest tests/test_gpu_ops.py -q -x -k "gemm" -p no:cacheprovider --tb=short 2>&1 | tail -n 25
run() {
  BENCH_GEMM_IMPL=$1 timeout 120 python bench.py --no-cpu-baseline --steps 40 --warmup 4 2>/dev/null | tail -n 1 | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('impl $1', d['value'], 'captions/s', d['ms_per_step'], 'ms/pass gemm avg us', round(1e3*d['roofline']['avg_launch_ms'],2), 'frac', d['roofline']['frac'], 'enc+prefill', d['phases_ms']['graph_encode_prefill_ms'], 'parity', (d.get('parity') or {}).get('identical'))"
}
for i in 1 2 3; do run 65545; run 9; done
