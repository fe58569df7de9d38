#!/bin/bash
# Round 3, fifth GPU call: beam-path kernels (vocabulary head walking its row blocks, wave-per-row candidate merge in the
# search step), the addln schedule on the round-2 GEMM kernel, full GPU suite
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD; TAG=${1:-r03_e}; T0=$(date +%s); t() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
rm -f gpurun_out/parity_measured.jsonl
t "full GPU suite"; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "not p9_persistent" > gpurun_out/${TAG}_pytest_gpu.txt 2>&1; echo "rc=$?"; tail -n 12 gpurun_out/${TAG}_pytest_gpu.txt | cut -c1-250
cp gpurun_out/parity_measured.jsonl gpurun_out/${TAG}_parity_measured.jsonl 2>/dev/null
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('parity') or {}; print(d['value'], 'captions/s', d['ms_per_step'], 'ms/pass | gemm frac', d['roofline']['frac'], '| enc+prefill', d['phases_ms']['graph_encode_prefill_ms'], 'decode', d['phases_ms']['graph_decode_ms'], 'step ms', d['roofline_decode']['avg_step_ms'], 'frac', d['roofline_decode']['frac'], '| parity', p.get('identical'), p.get('ok'), p.get('logit_err'))"; }
t "bench beam"; timeout 600 python bench.py --no-cpu-baseline --search beam --steps 12 --warmup 3 2> gpurun_out/${TAG}_beam.err | tee gpurun_out/${TAG}_bench_beam.json | line
for i in 1 2; do
  t "bench default ($i)"; timeout 600 python bench.py --no-cpu-baseline --steps 40 --warmup 8 2> gpurun_out/${TAG}_base.err | tee gpurun_out/${TAG}_bench_base_$i.json | line
  t "bench addln on p8 ($i)"; GITMI_ADDLN=1 timeout 600 python bench.py --no-cpu-baseline --steps 40 --warmup 8 2> gpurun_out/${TAG}_addln.err | tee gpurun_out/${TAG}_bench_addln_$i.json | line
done
cd /tmp; export TMPDIR=/tmp
t "rocprof beam solo"; timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_beam -o bench -- python $R/bench.py --no-cpu-baseline --search beam --contexts 1 --steps 6 --warmup 2 > $R/gpurun_out/${TAG}_beam_solo_bench.json 2> $R/gpurun_out/${TAG}_beam_solo.err; echo "rc=$?"
python $R/tools/rocprof_summary.py $R/gpurun_out/prof_beam/bench_results.db $R/gpurun_out/${TAG}_beam_solo_kernel_stats.txt > /dev/null; rm -rf $R/gpurun_out/prof_beam; head -n 14 $R/gpurun_out/${TAG}_beam_solo_kernel_stats.txt | cut -c1-200
cd $R
t done
