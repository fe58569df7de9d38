#!/bin/bash
# Round 3, seventh GPU call: wave-priority experiment -- the encoder GEMM's waves at s_setprio 2 / 3 (MFMA clusters) instead of
# 0 / 1, next to decode kernels at priority 0 (two builds of kernels_gemm10.o, interleaved bench runs)
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD; TAG=${1:-r03_g}; T0=$(date +%s); t() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
C=generativeimage2text_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result"
OBJS="$C/build/engine.o $C/build/kernels_gemm.o $C/build/kernels_gemm2.o $C/build/kernels_gemm3.o $C/build/kernels_gemm7.o $C/build/kernels_norm.o $C/build/kernels_attn.o $C/build/kernels_attn_decode.o $C/build/kernels_search.o $C/build/kernels_dgemm.o $C/build/kernels_preproc.o"
cp generativeimage2text_amd/libgitmi.so /tmp/libgitmi_prio0.so
t "build prio 2"; /opt/rocm/bin/hipcc $FLAGS -DP8_BASE_PRIO=2 -c $C/kernels_gemm10.hip -o /tmp/gemm10_p2.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/gemm10_p2.o -o /tmp/libgitmi_prio2.so; echo "rc=$?"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('parity') or {}; print(d['value'], 'captions/s', d['ms_per_step'], 'ms/pass | gemm frac', d['roofline']['frac'], '| enc+prefill', d['phases_ms']['graph_encode_prefill_ms'], 'decode', d['phases_ms']['graph_decode_ms'], 'lat', d['batch_latency_ms']['median'], '| parity', p.get('identical'), p.get('ok'))"; }
for i in 1 2 3; do
  for pr in 0 2; do
    cp /tmp/libgitmi_prio$pr.so generativeimage2text_amd/libgitmi.so
    t "bench GEMM base priority $pr ($i)"; timeout 600 python bench.py --no-cpu-baseline --steps 40 --warmup 8 2> gpurun_out/${TAG}_prio.err | tee gpurun_out/${TAG}_bench_prio${pr}_$i.json | line
  done
done
cp /tmp/libgitmi_prio0.so generativeimage2text_amd/libgitmi.so
t done
