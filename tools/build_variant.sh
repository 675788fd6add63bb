#!/bin/bash
# Build an alternative library of the march kernel with extra -D flags (A/B runs with tools/ab_march.py):
#   tools/build_variant.sh <name> [-DHAV_... ...]   ->  havatar_amd/lib/alt/libhavatar_<name>.so
# Only hav_render.hip is recompiled (HAV_FAST_BUILD: production kernels only); the other objects come from the regular build.
set -eu
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p havatar_amd/lib/alt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DHAV_FAST_BUILD -Wno-unused-value "$@" -c havatar_amd/csrc/hav_render.hip -o havatar_amd/lib/alt/hav_render_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o havatar_amd/lib/alt/libhavatar_$name.so havatar_amd/lib/alt/hav_render_$name.o havatar_amd/lib/hav_ops.o havatar_amd/lib/hav_train.o havatar_amd/lib/hav_mlp_train.o havatar_amd/lib/hav_conv.o
rm -f havatar_amd/lib/alt/hav_render_$name.o
echo havatar_amd/lib/alt/libhavatar_$name.so
