#!/bin/bash
# Round-end verification: full GPU parity suite, smoke, default bench line, rocprofv3 kernel stats of the same command,
# a solo (one context, eager) profile comparable with the roofline pass, and the other configs.
set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
R=$PWD
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.txt 2>&1; echo "rc=$?"; tail -4 gpurun_out/pytest_gpu.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "== bench"; timeout 900 python bench.py > gpurun_out/bench.txt 2>&1; echo "rc=$?"; tail -1 gpurun_out/bench.txt | cut -c1-400
cd /tmp; export TMPDIR=/tmp
echo "== rocprof (default bench command)"; timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o bench -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/rocprof.txt 2>&1; echo "rc=$?"
echo "== rocprof solo eager"; timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_solo -o bench -- python $R/bench.py --steps 5 --warmup 2 --contexts 1 --no-graph --no-cpu-baseline > $R/gpurun_out/rocprof_solo.txt 2>&1; echo "rc=$?"
cd $R
python tools/rocprof_summary.py gpurun_out/prof/bench_results.db gpurun_out/kernel_stats.txt > /dev/null; head -8 gpurun_out/kernel_stats.txt | cut -c1-200
python tools/rocprof_summary.py gpurun_out/prof_solo/bench_results.db gpurun_out/kernel_stats_solo.txt > /dev/null; head -8 gpurun_out/kernel_stats_solo.txt | cut -c1-200
tail -1 gpurun_out/rocprof_solo.txt > gpurun_out/bench_solo_eager.txt
run() { echo "== $*"; timeout 400 python bench.py --no-cpu-baseline --steps 12 --warmup 3 "$@" 2>&1 | grep -v amdgpu | tail -1 | tee -a gpurun_out/configs.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], 'ms/step', d['config']['workload'], d['roofline']['achieved'], 'TF', d['phases_ms'])"; }
rm -f gpurun_out/configs.jsonl
run --search beam
run --model GIT_LARGE_COCO --batch 32
run --model GIT_BASE_VATEX --frames 6 --batch 16
