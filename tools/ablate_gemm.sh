#!/bin/bash
# Builds variants of the library whose bf16 GEMM leaves one pipeline stage out (timing only - results are wrong):
# gamer_amd/lib/libgamer_hip_abl<k>.so for k in "$@"; tools/kbench_bf16.py --lib <path> times them.
set -e
cd "$(dirname "$0")/.."
python -m gamer_amd.build >/dev/null
SRC=${ABLATE_SRC:-gemm_bf16}          # gemm_bf16 (-DHB_ABLATE) or gemm (-DSP_ABLATE: the bf16-split fp32 GEMM)
MACRO=HB_ABLATE; [ "$SRC" = gemm ] && MACRO=SP_ABLATE
for k in "$@"; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++20 -Iinclude -Igamer_amd/csrc -D$MACRO=$k $ABLATE_EXTRA \
      -c gamer_amd/csrc/$SRC.hip -o /tmp/${SRC}_abl$k.o
  objs=$(ls gamer_amd/lib/*.o | grep -v "/$SRC.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gamer_amd/lib/libgamer_hip_abl$k.so $objs /tmp/${SRC}_abl$k.o
  echo built abl$k
done
