#!/bin/bash
# ONE collection script for a GPU box (replaces the per-call gpu_r03_* / gpu_r04_* scripts of earlier rounds):
#   bash tools/gpu_collect.sh TAG [section ...]        sections, default "pmc bench stats suite configs smoke":
#     pmc      rocprofv3 --pmc passes of tools/pmc_workload.py for the CURRENT csrc -> profiles/TAG_pmc_summary.tsv (bench.py reads
#              the newest summary whose csrc hash matches; copied to profiles/ FIRST so that `bench` below reports fresh traffic)
#     bench    the default line (python bench.py, cpu_baseline included) -> TAG_bench.json, and a 200-step line
#     stats    the default line under rocprofv3 --kernel-trace --stats -> TAG_default_kernel_stats.txt (+ solo policy, beam-4, GIT_LARGE, VATEX;
#              one row per (kernel, workgroup count), csrc hash in the header)
#     suite    python -m pytest tests -m gpu -> TAG_pytest_gpu.txt, TAG_parity_measured.jsonl
#     configs  bench lines of the other BASELINE configurations (beam 4, GIT_LARGE bs 32, VATEX 6 frames bs 16), bf16 and f16 builds
#     smoke    __graft_entry__.smoke()
#     attrib   tools/error_attribution.py on the benchmark's, the oracle's and the trained-statistics weights -> TAG_error_attribution_*.txt
# Everything lands in gpurun_out/ (merged back by gpurun); copy what is cited into profiles/.
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD; TAG=${1:-collect}; shift || true; SECTIONS=${*:-pmc bench stats suite configs smoke}
T0=$(date +%s); t() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
line() { python tools/bench_lines.py "$1" | cut -c1-230; }
for sec in $SECTIONS; do case $sec in
pmc)
  t pmc; bash tools/gpu_pmc.sh > gpurun_out/${TAG}_pmc.log 2>&1
  cp gpurun_out/pmc_summary.tsv gpurun_out/${TAG}_pmc_summary.tsv && cp gpurun_out/pmc_summary.tsv profiles/${TAG}_pmc_summary.tsv
  rm -rf gpurun_out/pmc; grep csrc_sha gpurun_out/${TAG}_pmc_summary.tsv | cut -c1-120 ;;
bench)
  t "default line"; timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "rc=$?"; line gpurun_out/${TAG}_bench.json
  t "200 steps"; timeout 200 python bench.py --no-cpu-baseline --steps 200 --warmup 8 > gpurun_out/${TAG}_bench_200steps.json 2> /dev/null; line gpurun_out/${TAG}_bench_200steps.json
  t "one context, solo policy"; timeout 200 python bench.py --no-cpu-baseline --contexts 1 --steps 10 --warmup 2 > gpurun_out/${TAG}_solo_bench.json 2> /dev/null; line gpurun_out/${TAG}_solo_bench.json ;;
stats)
  t "rocprofv3 --kernel-trace --stats: default line"
  ( cd /tmp; export TMPDIR=/tmp
    for v in "default:" "solo:--contexts 1 --steps 10 --warmup 2" "beam4:--search beam --contexts 1 --steps 10 --warmup 2" \
             "large_b32:--model GIT_LARGE --batch 32 --contexts 1 --steps 8 --warmup 2" \
             "vatex_b16:--model GIT_BASE_VATEX --frames 6 --batch 16 --contexts 1 --steps 8 --warmup 2"; do
      n=${v%%:*}; a=${v#*:}
      timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$n -o bench -- python $R/bench.py --no-cpu-baseline --no-alt-precision --no-teacher-forced $a > $R/gpurun_out/${TAG}_${n}_rocprof_bench.json 2> $R/gpurun_out/${TAG}_${n}_rocprof.err
      python $R/tools/rocprof_summary.py $R/gpurun_out/prof_$n/bench_results.db $R/gpurun_out/${TAG}_${n}_kernel_stats.txt > /dev/null
      rm -rf $R/gpurun_out/prof_$n; head -n 12 $R/gpurun_out/${TAG}_${n}_kernel_stats.txt | cut -c1-200
    done ) ;;
suite)
  rm -f gpurun_out/parity_measured.jsonl
  t "pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/${TAG}_pytest_gpu.txt 2>&1; echo "rc=$?"; tail -n 3 gpurun_out/${TAG}_pytest_gpu.txt | cut -c1-250
  cp gpurun_out/parity_measured.jsonl gpurun_out/${TAG}_parity_measured.jsonl 2>/dev/null ;;
configs)
  for c in "beam4:--search beam" "large_b32:--model GIT_LARGE --batch 32" "vatex_b16:--model GIT_BASE_VATEX --frames 6 --batch 16"; do
    n=${c%%:*}; a=${c#*:}
    for prec in bf16 f16; do
      t "$n $prec"; timeout 300 python bench.py --no-cpu-baseline --precision $prec $a > gpurun_out/${TAG}_${n}_${prec}_bench.json 2> /dev/null; line gpurun_out/${TAG}_${n}_${prec}_bench.json
    done
  done
  t "base f16"; timeout 300 python bench.py --no-cpu-baseline --precision f16 > gpurun_out/${TAG}_base_f16_bench.json 2> /dev/null; line gpurun_out/${TAG}_base_f16_bench.json ;;
smoke)
  t smoke; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.txt 2>&1; echo "rc=$?"; tail -n 2 gpurun_out/${TAG}_smoke.txt | cut -c1-200 ;;
attrib)
  for wt in bench oracle trained; do
    t "error attribution: $wt weights"; timeout 300 python tools/error_attribution.py --weights $wt --out gpurun_out/${TAG}_error_attribution_$wt.txt > gpurun_out/${TAG}_attrib_$wt.log 2>&1; echo "rc=$?"
    cat gpurun_out/${TAG}_error_attribution_$wt.txt | cut -c1-170
  done ;;
*) echo "unknown section $sec" ;;
esac; done
t done
