#!/bin/bash
# Round-end verification and profile collection (one gpurun call): full GPU parity suite, smoke, the default bench
# line, rocprofv3 kernel stats of the same command and of a solo (one context) graph-replay run, PMC passes, the other
# BASELINE configs with their own kernel stats, CPU-baseline thread sweep.  Summaries land in gpurun_out/ (copy the
# ones to keep into profiles/).
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
TAG=${1:-final}
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/${TAG}_pytest_gpu.txt 2>&1; echo "rc=$?"; tail -n 3 gpurun_out/${TAG}_pytest_gpu.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2
echo "== bench"; timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "rc=$?"; tail -n 1 gpurun_out/${TAG}_bench.json | cut -c1-300
cd /tmp; export TMPDIR=/tmp
prof() {  # name, bench args...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$name -o bench -- python $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/${TAG}_${name}_bench.json 2> $R/gpurun_out/${TAG}_${name}.err; echo "rocprof $name rc=$?"
  python $R/tools/rocprof_summary.py $R/gpurun_out/prof_$name/bench_results.db $R/gpurun_out/${TAG}_${name}_kernel_stats.txt > /dev/null
  if [ "$name" = solo_graph ]; then cp $R/gpurun_out/prof_$name/bench_results.db /tmp/solo_results.db; fi
  if [ "$name" = default ]; then   # where the wall time of the mixed schedule goes (concurrency, start delays, GEMM inflation vs solo)
    python $R/tools/mix_timeline.py $R/gpurun_out/prof_$name/bench_results.db --solo /tmp/solo_results.db --out $R/gpurun_out/${TAG}_mix_timeline.txt > /dev/null
  fi
  rm -rf $R/gpurun_out/prof_$name
  tail -n 1 $R/gpurun_out/${TAG}_${name}_bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], 'ms/step', d['config']['workload'], d['roofline']['achieved'], 'TF', d.get('roofline_decode',{}).get('avg_step_ms'))"
}
prof solo_graph --contexts 1 --steps 8 --warmup 2
prof default --steps 40 --warmup 8
prof beam4 --search beam --steps 12 --warmup 3
prof large_b32 --model GIT_LARGE_COCO --batch 32 --steps 12 --warmup 3
prof vatex_b16 --model GIT_BASE_VATEX --frames 6 --batch 16 --steps 12 --warmup 3
cd $R
echo "== pmc"; bash tools/gpu_pmc.sh > gpurun_out/${TAG}_pmc.log 2>&1; cp gpurun_out/pmc_summary.tsv gpurun_out/${TAG}_pmc_summary.tsv; head -n 12 gpurun_out/${TAG}_pmc_summary.tsv | cut -c1-240
rm -rf gpurun_out/pmc
echo "== cpu sweep"; timeout 900 python bench.py --cpu-sweep > gpurun_out/${TAG}_cpu_sweep.json 2>&1; tail -n 1 gpurun_out/${TAG}_cpu_sweep.json | cut -c1-600
