#!/bin/bash
# Round 3, first GPU call: the whole GPU suite (incl. the un-gated search-method tests, repetition penalty, the fp16
# residual-stream tests), the error-attribution table, an interleaved A/B of the fp16 residual stream, and the baseline
# kernel statistics of this round (solo graph replay + the default mixed schedule).
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD; TAG=${1:-r03_a}
rm -f gpurun_out/parity_measured.jsonl
T0=$(date +%s); t() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
t "pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/${TAG}_pytest_gpu.txt 2>&1; echo "rc=$?"; tail -n 25 gpurun_out/${TAG}_pytest_gpu.txt | cut -c1-300
cp gpurun_out/parity_measured.jsonl gpurun_out/${TAG}_parity_measured.jsonl 2>/dev/null
t "error attribution"; timeout 600 python tools/error_attribution.py --out gpurun_out/${TAG}_error_attribution.txt 2>&1 | tail -n 12
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('parity') or {}; print(d['value'], 'captions/s', d['ms_per_step'], 'ms/pass | gemm frac', d['roofline']['frac'], 'avg launch ms', d['roofline']['avg_launch_ms'], '| enc+prefill', d['phases_ms']['graph_encode_prefill_ms'], 'decode', d['phases_ms']['graph_decode_ms'], '| parity', p.get('identical'), p.get('ok'), p.get('logit_err'))"; }
for i in 1 2; do
  for f in 0 1; do
    t "bench STREAM_F16=$f ($i)"; GITMI_STREAM_F16=$f timeout 600 python bench.py --no-cpu-baseline --steps 40 --warmup 8 2> gpurun_out/${TAG}_f16_${f}.err | tee gpurun_out/${TAG}_bench_f16_${f}_$i.json | line
  done
done
cd /tmp; export TMPDIR=/tmp
prof() {  # name, bench args...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$name -o bench -- python $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/${TAG}_${name}_bench.json 2> $R/gpurun_out/${TAG}_${name}.err; echo "rocprof $name rc=$?"
  python $R/tools/rocprof_summary.py $R/gpurun_out/prof_$name/bench_results.db $R/gpurun_out/${TAG}_${name}_kernel_stats.txt > /dev/null
  rm -rf $R/gpurun_out/prof_$name
  head -n 14 $R/gpurun_out/${TAG}_${name}_kernel_stats.txt | cut -c1-200
}
t "rocprof solo graph"; prof solo_graph --contexts 1 --steps 8 --warmup 2
t "rocprof default"; prof default --steps 40 --warmup 8
cd $R
t "default bench line (with cpu baseline)"; timeout 900 python bench.py 2> gpurun_out/${TAG}_bench.err | tee gpurun_out/${TAG}_bench.json | cut -c1-400
t done
