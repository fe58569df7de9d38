#!/bin/bash
# builds the standalone GEMM probes next to their sources (binaries are git-ignored; they travel with gpurun)
set -e; cd "$(dirname "$0")"
for f in *.hip; do
  /opt/rocm/bin/hipcc -DGITMI_PROBE --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-unused-result -I ../../generativeimage2text_amd/csrc "$f" -o "${f%.hip}"
done
