#!/bin/bash
# Same-box A/B of two builds of libgamer_hip.so (boxes differ by +-3 %, so A and B must run on one box):
#   1. build variant A, cp gamer_amd/lib/libgamer_hip.so gamer_amd/lib/old.so ; same for B -> new.so
#   2. gpurun -- bash tools/ab_libs.sh attn|gemm|elem [B]      (old, new, old, new)
# The library in place afterwards is new.so.
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
WHAT=${1:-gemm}; B=${2:-1024}
for r in 1 2; do
  for v in old new; do
    cp gamer_amd/lib/$v.so gamer_amd/lib/libgamer_hip.so
    echo "== $v"
    timeout 200 python tools/kbench.py $WHAT --B $B 2>&1 | grep -E "^gemm|^attn|GB/s"
  done
done
cp gamer_amd/lib/new.so gamer_amd/lib/libgamer_hip.so
