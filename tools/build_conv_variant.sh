#!/bin/bash
# Alternative library with extra -D flags for hav_conv.hip only (timing experiments): tools/build_conv_variant.sh <name> [-D...]
set -eu
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p havatar_amd/lib/alt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c havatar_amd/csrc/hav_conv.hip -o havatar_amd/lib/alt/hav_conv_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o havatar_amd/lib/alt/libhavatar_$name.so havatar_amd/lib/alt/hav_conv_$name.o havatar_amd/lib/hav_ops.o havatar_amd/lib/hav_train.o havatar_amd/lib/hav_mlp_train.o havatar_amd/lib/hav_render.o
rm -f havatar_amd/lib/alt/hav_conv_$name.o
echo havatar_amd/lib/alt/libhavatar_$name.so
