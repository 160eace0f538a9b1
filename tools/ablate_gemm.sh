#!/bin/bash
# Builds variants of the library whose bf16 GEMM leaves one pipeline stage out (timing only - results are wrong):
# gamer_amd/lib/libgamer_hip_abl<k>.so for k in "$@"; tools/kbench_bf16.py --lib <path> times them.
set -e
cd "$(dirname "$0")/.."
python -m gamer_amd.build >/dev/null
for k in "$@"; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++20 -Iinclude -Igamer_amd/csrc -DHB_ABLATE=$k \
      -c gamer_amd/csrc/gemm_bf16.hip -o /tmp/gemm_bf16_abl$k.o
  objs=$(ls gamer_amd/lib/*.o | grep -v gemm_bf16)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gamer_amd/lib/libgamer_hip_abl$k.so $objs /tmp/gemm_bf16_abl$k.o
  echo built abl$k
done
