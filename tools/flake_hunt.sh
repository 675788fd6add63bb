#!/bin/bash
# tools/flake_hunt.sh N "ENV=1 ENV2=0" ... : run the GPU training-CLI test N times per environment, count failures, keep the output of failures
N=$1; shift
mkdir -p gpurun_out/flake
k=0
for e in "$@"; do
  fails=0
  for i in $(seq 1 $N); do
    out=$(env $e python -m pytest tests/test_harness.py -m gpu -q -x -s -k "graph_mode_runs" 2>&1)
    if echo "$out" | grep -q "1 failed"; then
      fails=$((fails+1)); k=$((k+1))
      echo "$out" > gpurun_out/flake/fail_$k.txt
      echo "$out" | grep "nan-trace" | cut -c1-1200
    fi
  done
  echo "== $e : $fails / $N failed"
done
