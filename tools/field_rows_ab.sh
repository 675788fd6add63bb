#!/bin/bash
# per-kernel time of the field-input scatter inside the training step (tools/bench_train.py), ray-major runs against rows of rays
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/field_rows_ab
rm -rf $OUT; mkdir -p $OUT
for r in 0 1 0 1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p$r -o p -- env HAVATAR_FIELD_ROWS=$r BENCH_TRAIN_NO_BREAKDOWN=1 python tools/bench_train.py > $OUT/log$r.txt 2>&1
  f=$(find $OUT/p$r -name "*kernel_stats.csv" | head -1)
  echo "HAVATAR_FIELD_ROWS=$r: $(grep 'train step' $OUT/log$r.txt)"
  grep -E "field_inputs_kernel<2|field_inputs_kernel<0|field_inputs_run_kernel|composite_kernel<2|mlp_bwd_data" "$f" | cut -c1-160
  rm -rf $OUT/p$r
done 2>&1 | tee $OUT/field_rows_ab.txt
