#!/bin/bash
# Round 3 collection in one call: full GPU suite, smoke, PMC passes for the current csrc (copied to profiles/ so that the
# bench line of this very run carries fresh `traffic`), the default bench line (with the CPU baseline), rocprofv3 kernel
# statistics of the solo graph replay and of the default mixed schedule, bench lines of the other BASELINE configs in
# both 16-bit operand builds, the default line without the serving policy and with decode groups of 2.
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD; TAG=${1:-r03_z}; T0=$(date +%s); t() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
rm -f gpurun_out/parity_measured.jsonl
t "pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/${TAG}_pytest_gpu.txt 2>&1; echo "rc=$?"; tail -n 6 gpurun_out/${TAG}_pytest_gpu.txt | cut -c1-250
cp gpurun_out/parity_measured.jsonl gpurun_out/${TAG}_parity_measured.jsonl 2>/dev/null
t "smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
t "pmc"; bash tools/gpu_pmc.sh > gpurun_out/${TAG}_pmc.log 2>&1; cp gpurun_out/pmc_summary.tsv gpurun_out/${TAG}_pmc_summary.tsv; cp gpurun_out/pmc_summary.tsv profiles/${TAG}_pmc_summary.tsv; rm -rf gpurun_out/pmc; head -n 14 gpurun_out/${TAG}_pmc_summary.tsv | cut -c1-200
t "default bench"; timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "rc=$?"; tail -n 1 gpurun_out/${TAG}_bench.json | cut -c1-400
t "default bench, 200 steps"; timeout 900 python bench.py --no-cpu-baseline --steps 200 --warmup 8 2> /dev/null | tail -n 1 > gpurun_out/${TAG}_bench_200steps.json; python -c "import json; d=json.load(open('gpurun_out/${TAG}_bench_200steps.json')); print(d['value'], d['ms_per_step'])"
cd /tmp; export TMPDIR=/tmp
prof() {
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$name -o bench -- python $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/${TAG}_${name}_bench.json 2> $R/gpurun_out/${TAG}_${name}.err; echo "rocprof $name rc=$?"
  python $R/tools/rocprof_summary.py $R/gpurun_out/prof_$name/bench_results.db $R/gpurun_out/${TAG}_${name}_kernel_stats.txt > /dev/null
  rm -rf $R/gpurun_out/prof_$name
}
t "rocprof solo graph"; prof solo_graph --contexts 1 --steps 8 --warmup 2; head -n 12 $R/gpurun_out/${TAG}_solo_graph_kernel_stats.txt | cut -c1-180
t "rocprof default"; prof default --steps 40 --warmup 8
t "rocprof beam solo"; prof beam4_solo --search beam --contexts 1 --steps 6 --warmup 2
t "rocprof large solo"; prof large_solo --model GIT_LARGE_COCO --batch 32 --contexts 1 --steps 6 --warmup 2
cd $R
run() { local name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --steps 20 --warmup 4 "$@" 2> gpurun_out/${TAG}_${name}.err | tail -n 1 > gpurun_out/${TAG}_${name}_bench.json; python -c "import json; d=json.load(open('gpurun_out/${TAG}_${name}_bench.json')); p=d.get('parity') or {}; print('$name', d['dtype'], d['value'], d['ms_per_step'], 'ms | gemm', d['roofline']['frac'], 'decode frac', d['roofline_decode']['frac'], 'step', d['roofline_decode']['avg_step_ms'], '| parity', p.get('identical'), '/', p.get('rows'), p.get('ok'))"; }
run base_solo_policy --solo-policy
run base_decode_group2 --decode-group 2
for pr in bf16 f16; do
  run base_${pr} --precision $pr
  run beam4_${pr} --search beam --precision $pr
  run large_b32_${pr} --model GIT_LARGE_COCO --batch 32 --precision $pr
  run vatex_b16_${pr} --model GIT_BASE_VATEX --frames 6 --batch 16 --precision $pr
done
t done
