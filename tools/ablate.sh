#!/bin/bash
# timing ablations of the march kernel (HAV_ABLATE bit mask; results are wrong, only kernel_ms matters)
cd "$GRAFT_REPO_ROOT"
for m in 0 1 2 4 8 16 32 64 128 256 48 63 511; do
  HAV_ABLATE=$m python bench.py --steps 6 --warmup 2 --no-cpu-baseline --perturb 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ablate=$m kernel_ms', d['roofline']['kernel_ms'])"
done
