set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
rm -f gpurun_out/parity_measured.jsonl
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_policy.py -m gpu -q -x --tb=short -p no:cacheprovider -k "teacher_forced or policy" > gpurun_out/r05_a_pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 5 gpurun_out/r05_a_pytest.txt | cut -c1-300
cp gpurun_out/parity_measured.jsonl gpurun_out/r05_a_parity_measured.jsonl
timeout 600 python bench.py > gpurun_out/r05_a_bench.json 2> gpurun_out/r05_a_bench.err; echo "bench rc=$?"; tail -c 1500 gpurun_out/r05_a_bench.err
python tools/bench_lines.py gpurun_out/r05_a_bench.json | cut -c1-400
