#!/bin/bash
# A second build of libgamer_hip.so with extra -D flags on ONE source file (same-box A/B runs: GAMER_LIB_PATH=<that .so>):
#   bash tools/build_variant.sh noguard gemm.hip -DGAMER_SPLIT3_GUARD_BUILD=0   ->  tools/_ab/noguard.so
# Needs the object files of a normal build (python -m gamer_amd.build) in gamer_amd/lib/.
set -e
cd "$(dirname "$0")/.."
NAME=$1; SRC=$2; shift 2
L=gamer_amd/lib
mkdir -p tools/_ab
OBJ=tools/_ab/${SRC%.hip}.$NAME.o
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++20 -Wno-unused-function "$@" -c gamer_amd/csrc/$SRC -o $OBJ
OBJS=""
for o in prep inject elementwise gemm gemm_as gemm_wg gemm_os gemm_bf16 attention attention_split attention_res attention_bf16 optim decode modules; do
  if [ "$o.hip" == "$SRC" ]; then OBJS="$OBJS $OBJ"; else OBJS="$OBJS $L/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_ab/$NAME.so $OBJS
echo "built tools/_ab/$NAME.so"
