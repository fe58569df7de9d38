set -u; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
AB_STEPS=40 AB_WARMUP=8 bash tools/gpu_ab.sh r05_m 2 "walk60:GITMI_VOCAB_WGS=60 -- --search beam" "walk120:GITMI_VOCAB_WGS=120 -- --search beam" "block239:GITMI_VOCAB_WGS=0 -- --search beam"
