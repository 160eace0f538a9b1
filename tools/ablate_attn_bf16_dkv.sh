#!/bin/bash
# Host: timing-only builds of attn_bwd_dkv_b_kernel (AB_ABLATE bits, csrc/attention_bf16.hip) -> tools/_ab/dkvb_<bits>.so;
# on the GPU box: for f in tools/_ab/dkvb_*.so; do GAMER_LIB_PATH=$f python tools/time_attn_bf16_bwd.py; done
set -e
cd "$(dirname "$0")/.."
for bits in "$@"; do
  bash tools/build_variant.sh dkvb_$bits attention_bf16.hip -DAB_ABLATE_DKV=$bits &
done
wait
