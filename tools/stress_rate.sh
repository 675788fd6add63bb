#!/bin/bash
# failure rate of the production kernel's bitwise determinism for several library builds: stress_rate.sh lib1.so lib2.so ...  (LAUNCHES each)
cd "$GRAFT_REPO_ROOT"
for l in "$@"; do
  echo -n "$l: "
  HAVATAR_LIB=$PWD/$l LAUNCHES=${LAUNCHES:-1000} timeout 900 python tools/stress_diag.py 2>&1 | grep "differing outputs" | tr '\n' ' '
  echo
done
