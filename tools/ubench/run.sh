#!/bin/bash
# build + run the microbenchmarks on the GPU box: MFMA / VALU overlap, per-wave issue cost and the wave -> SIMD map (3.11),
# what v_permlane32_swap_b32 does (3.12), whether a matrix instruction reads its operands after issue (3.5: it does not)
cd "$GRAFT_REPO_ROOT/tools/ubench"
for f in mfma_valu_overlap mfma_fill mfma_dep issue_cost simd_map permlane32 mfma_war sincos_acc; do
  hipcc --offload-arch=gfx950 -O3 -o /tmp/$f $f.hip 2>/dev/null && echo "== $f" && /tmp/$f
done
