set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_ln_fold.py tests/test_gpu_policy.py tests/test_gpu_ops.py -q -x --tb=short -p no:cacheprovider 2>&1 | tail -15
F="--no-cpu-baseline --no-alt-precision --no-teacher-forced --no-other-configs"
for i in 1 2; do
 for v in "fold:" "nofold:--no-ln-fold"; do n=${v%%:*}; a=${v#*:}
  timeout 200 python bench.py $F $a > gpurun_out/ab_${n}_$i.json 2>gpurun_out/ab_${n}_$i.err; python tools/bench_lines.py gpurun_out/ab_${n}_$i.json | cut -c1-230
 done
done
for v in "fold:" "nofold:--no-ln-fold"; do n=${v%%:*}; a=${v#*:}
  timeout 200 python bench.py $F --contexts 1 --steps 10 --warmup 2 $a > gpurun_out/ab_solo_${n}.json 2>/dev/null; python tools/bench_lines.py gpurun_out/ab_solo_${n}.json | cut -c1-230
done
