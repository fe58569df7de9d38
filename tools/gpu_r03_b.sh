#!/bin/bash
# Round 3, second GPU call: the persistent GEMM (kernels_gemm11.hip) and the addln schedule -- unit tests, isolated A/B
# against the round-2 kernel, the engine-level parity cases under the new schedule, interleaved A/B of the whole bench,
# then the full GPU suite (fixed bounds).
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD; TAG=${1:-r03_b}
rm -f gpurun_out/parity_measured.jsonl
T0=$(date +%s); t() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
t "p9 / add_ln unit tests"; timeout 900 python -m pytest tests/test_gpu_ops.py -q --tb=short -p no:cacheprovider -x -k "p9 or add_layernorm" > gpurun_out/${TAG}_pytest_p9.txt 2>&1; echo "rc=$?"; tail -n 12 gpurun_out/${TAG}_pytest_p9.txt | cut -c1-250
t "isolated GEMM A/B"; timeout 600 python tools/gemm_p9_bench.py 2>&1 | grep -v amdgpu | tee gpurun_out/${TAG}_gemm_p9_bench.txt
t "addln engine tests"; timeout 900 python -m pytest tests/test_gpu_addln.py -q --tb=short -p no:cacheprovider > gpurun_out/${TAG}_pytest_addln.txt 2>&1; echo "rc=$?"; tail -n 12 gpurun_out/${TAG}_pytest_addln.txt | cut -c1-250
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('parity') or {}; print(d['value'], 'captions/s', d['ms_per_step'], 'ms/pass | gemm frac', d['roofline']['frac'], 'avg launch ms', d['roofline']['avg_launch_ms'], '| enc+prefill', d['phases_ms']['graph_encode_prefill_ms'], 'decode', d['phases_ms']['graph_decode_ms'], '| parity', p.get('identical'), p.get('ok'), p.get('logit_err'))"; }
for i in 1 2; do
  t "bench default ($i)"; timeout 600 python bench.py --no-cpu-baseline --steps 40 --warmup 8 2> gpurun_out/${TAG}_ab_base.err | tee gpurun_out/${TAG}_bench_base_$i.json | line
  t "bench p9 only ($i)"; GITMI_GEMM_IMPL=11 timeout 600 python bench.py --no-cpu-baseline --steps 40 --warmup 8 2> gpurun_out/${TAG}_ab_p9.err | tee gpurun_out/${TAG}_bench_p9_$i.json | line
  t "bench addln + p9 ($i)"; GITMI_ADDLN=1 GITMI_GEMM_IMPL=11 timeout 600 python bench.py --no-cpu-baseline --steps 40 --warmup 8 2> gpurun_out/${TAG}_ab_addln.err | tee gpurun_out/${TAG}_bench_addln_p9_$i.json | line
  t "bench addln + p9 ring8 ($i)"; GITMI_ADDLN=1 GITMI_GEMM_IMPL=$((11 | (512 << 8))) timeout 600 python bench.py --no-cpu-baseline --steps 40 --warmup 8 2> gpurun_out/${TAG}_ab_addln8.err | tee gpurun_out/${TAG}_bench_addln_p9ring8_$i.json | line
done
t "beam parity line"; timeout 600 python bench.py --no-cpu-baseline --search beam --steps 12 --warmup 3 2> gpurun_out/${TAG}_beam.err | tee gpurun_out/${TAG}_bench_beam.json | line
cd /tmp; export TMPDIR=/tmp
t "rocprof addln + p9 solo"; GITMI_ADDLN=1 GITMI_GEMM_IMPL=11 timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_solo -o bench -- python $R/bench.py --no-cpu-baseline --contexts 1 --steps 8 --warmup 2 > $R/gpurun_out/${TAG}_addln_solo_bench.json 2> $R/gpurun_out/${TAG}_addln_solo.err; echo "rc=$?"
python $R/tools/rocprof_summary.py $R/gpurun_out/prof_solo/bench_results.db $R/gpurun_out/${TAG}_addln_solo_kernel_stats.txt > /dev/null; rm -rf $R/gpurun_out/prof_solo; head -n 16 $R/gpurun_out/${TAG}_addln_solo_kernel_stats.txt | cut -c1-200
cd $R
t "full GPU suite"; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --deselect tests/test_gpu_addln.py -k "not p9_persistent" > gpurun_out/${TAG}_pytest_gpu.txt 2>&1; echo "rc=$?"; tail -n 15 gpurun_out/${TAG}_pytest_gpu.txt | cut -c1-250
cp gpurun_out/parity_measured.jsonl gpurun_out/${TAG}_parity_measured.jsonl 2>/dev/null
t done
