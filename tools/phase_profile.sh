#!/bin/bash
# Build the march kernel with per-phase s_memtime instrumentation into an ALTERNATIVE library and print the breakdown.
# Run on the GPU box: gpurun -- 'bash tools/phase_profile.sh'   (the alternative .so must be built before, see below)
set -eu
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ALT=havatar_amd/lib/alt/libhavatar_hip_prof.so
if [ ! -f "$ALT" ] || [ havatar_amd/csrc/hav_render.hip -nt "$ALT" ] || [ havatar_amd/csrc/hav_ops.hip -nt "$ALT" ] || [ havatar_amd/csrc/hav_train.hip -nt "$ALT" ]; then
  mkdir -p havatar_amd/lib/alt
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DHAV_PROFILE ${EXTRA:-} -c havatar_amd/csrc/hav_render.hip -o havatar_amd/lib/alt/hav_render_prof.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c havatar_amd/csrc/hav_ops.hip -o havatar_amd/lib/alt/hav_ops.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c havatar_amd/csrc/hav_train.hip -o havatar_amd/lib/alt/hav_train.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ALT" havatar_amd/lib/alt/hav_render_prof.o havatar_amd/lib/alt/hav_ops.o havatar_amd/lib/alt/hav_train.o
fi
if [ "${BUILD_ONLY:-0}" = "1" ]; then exit 0; fi
for P in 1 0; do HAVATAR_LIB=$PWD/$ALT PERTURB=$P python tools/phase_profile.py; done
