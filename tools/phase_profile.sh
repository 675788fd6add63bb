#!/bin/bash
# Build the march kernel with per-phase s_memtime instrumentation into an ALTERNATIVE library and print the breakdown.
# Run on the GPU box: gpurun -- 'bash tools/phase_profile.sh'   (the alternative .so must be built before, see below)
set -eu
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ALT=havatar_amd/lib/alt/libhavatar_hip_prof.so
if [ ! -f "$ALT" ] || [ havatar_amd/csrc/hav_render.hip -nt "$ALT" ]; then
  mkdir -p havatar_amd/lib/alt
  [ -f havatar_amd/lib/hav_ops.o ] || python -m havatar_amd.build
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DHAV_PROFILE -DHAV_FAST_BUILD ${EXTRA:-} -c havatar_amd/csrc/hav_render.hip -o havatar_amd/lib/alt/hav_render_prof.o
  # every other object comes from the regular build
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ALT" havatar_amd/lib/alt/hav_render_prof.o havatar_amd/lib/hav_ops.o havatar_amd/lib/hav_train.o havatar_amd/lib/hav_mlp_train.o havatar_amd/lib/hav_conv.o
fi
if [ "${BUILD_ONLY:-0}" = "1" ]; then exit 0; fi
for P in ${PERTURBS:-1 0}; do HAVATAR_LIB=$PWD/$ALT PERTURB=$P python tools/phase_profile.py; done
