#!/bin/bash
# Socket power / clocks while the march kernel replays back to back, per HAV_ABLATE setting (timing-only ablations of an alternative
# library): which part of the kernel the package-power limiter is paying for.  usage (GPU box): tools/power_probe.sh lib.so "0 1 52 ..."
cd "$GRAFT_REPO_ROOT"
LIB=$1; shift
for A in $1; do
  (HAV_ABLATE=$A HAVATAR_LIB=$PWD/$LIB HAVATAR_MLP=${MODE:-half} LAUNCHES=${LAUNCHES:-1100} python tools/march_once.py > /dev/null 2>&1 &)
  sleep 6.5
  P=""; C=""
  for i in 1 2 3; do
    M=$(rocm-smi --showmetrics 2>/dev/null)
    P="$P $(echo "$M" | grep -m1 current_socket_power | grep -oE '[0-9]+$')"
    C="$C $(echo "$M" | grep -m1 'current_gfxclk ' | grep -oE '[0-9]+$')"
    sleep 0.3
  done
  T=$(amd-smi metric --throttle 2>/dev/null | grep -m1 PPT_VIOLATION_ACTIVITY | grep -oE '[0-9]+ %')
  echo "HAV_ABLATE=$A  socket W:$P  gfxclk MHz:$C  PPT activity: $T"
  wait; sleep 4
done
