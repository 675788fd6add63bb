#!/bin/bash
# Per-phase cycle breakdown of conv3x3_split_kernel's main loop (alternative build with -DCV_PROFILE, s_memtime per phase and wave).
# Build here (no GPU needed): BUILD_ONLY=1 bash tools/conv_phase.sh ; run on the GPU box: gpurun -- 'bash tools/conv_phase.sh'
set -eu
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ALT=havatar_amd/lib/alt/libhavatar_hip_cvprof.so
if [ ! -f "$ALT" ] || [ havatar_amd/csrc/hav_conv.hip -nt "$ALT" ]; then
  mkdir -p havatar_amd/lib/alt
  for f in hav_ops hav_render hav_train hav_mlp_train; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c havatar_amd/csrc/$f.hip -o havatar_amd/lib/alt/$f.o &
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DCV_PROFILE -c havatar_amd/csrc/hav_conv.hip -o havatar_amd/lib/alt/hav_conv_prof.o
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ALT" havatar_amd/lib/alt/hav_ops.o havatar_amd/lib/alt/hav_render.o havatar_amd/lib/alt/hav_train.o havatar_amd/lib/alt/hav_mlp_train.o havatar_amd/lib/alt/hav_conv_prof.o
fi
if [ "${BUILD_ONLY:-0}" = "1" ]; then exit 0; fi
HAVATAR_LIB=$PWD/$ALT python tools/conv_phase.py
