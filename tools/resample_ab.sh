#!/bin/bash
# the training step (tools/bench_train.py) with the importance resampling as the ATen statement and as hav_resample_depths, twice each
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/resample_ab
rm -rf $OUT; mkdir -p $OUT
{
  python tools/bench_resample.py
  for r in aten hip aten hip; do
    HAVATAR_RESAMPLE=$r BENCH_TRAIN_NO_BREAKDOWN=1 python tools/bench_train.py > $OUT/log_$r.txt 2>&1
    echo "HAVATAR_RESAMPLE=$r: $(grep 'train step' $OUT/log_$r.txt | tr '\n' ' ')"
  done
} 2>&1 | tee $OUT/resample_ab.txt
