#!/bin/bash
# build + run the MFMA/VALU overlap microbenchmarks on the GPU box (evidence for DESIGN.md section 3.3)
cd "$GRAFT_REPO_ROOT/tools/ubench"
for f in mfma_valu_overlap mfma_fill mfma_dep; do
  hipcc --offload-arch=gfx950 -O3 -o /tmp/$f $f.hip 2>/dev/null && echo "== $f" && /tmp/$f
done
