# Timing-only builds of csrc/attention_split.hip that leave one ingredient out (SPA_ABLATE bits, see the source): each is
# compiled alone and linked with the other objects of the current build into tools/_ab/libgamer_spa<bits>.so.
# usage (build container): bash tools/ablate_attn_split.sh 1 2 3 4 8 16 ...
cd "$(dirname "$0")/.."
L=gamer_amd/lib
mkdir -p tools/_ab
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++20 -Wno-unused-function -DSPA_ABLATE=$v -c gamer_amd/csrc/attention_split.hip -o tools/_ab/attention_split_spa$v.o || exit 1
  objs=$(ls $L/*.o | grep -v attention_split.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_ab/libgamer_spa$v.so $objs tools/_ab/attention_split_spa$v.o || exit 1
  rm tools/_ab/attention_split_spa$v.o
  echo built tools/_ab/libgamer_spa$v.so
done
