set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
rm -f gpurun_out/parity_measured.jsonl
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity.py tests/test_gpu_policy.py -m gpu -q --tb=short -p no:cacheprovider -k "vocab or topm or beam or scripted or keep_best or sampling or trie or full_batch or wide or policy or teacher" > gpurun_out/r05_o_pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 4 gpurun_out/r05_o_pytest.txt | cut -c1-300
cp gpurun_out/parity_measured.jsonl gpurun_out/r05_o_parity_measured.jsonl
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --search beam > gpurun_out/r05_o_beam4_$i.json 2>/dev/null; python tools/bench_lines.py gpurun_out/r05_o_beam4_$i.json | cut -c1-230; done
timeout 300 python bench.py --no-cpu-baseline --no-alt-precision > gpurun_out/r05_o_default.json 2>/dev/null; python tools/bench_lines.py gpurun_out/r05_o_default.json | cut -c1-230
