#!/bin/bash
# Effective shader clock of the march kernel per library build: GRBM_GUI_ACTIVE / 8 XCDs / kernel duration (rocprofv3, one PMC pass with
# the kernel trace), over a sustained run of LAUNCHES launches; plus rocm-smi power / clock samples during an un-profiled loop.
# usage (GPU box): tools/clock_probe.sh libA.so libB.so ...
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  d=$(mktemp -d /tmp/clk_XXXX)
  HAVATAR_LIB=$PWD/$lib LAUNCHES=${LAUNCHES:-40} timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $d -o p -- python tools/march_once.py > $d/log 2>&1
  python3 - "$d" "$lib" <<'PY'
import csv, glob, sys, collections
d, lib = sys.argv[1], sys.argv[2]
dur = {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "march_blk" in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
cnt = collections.defaultdict(dict)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "march_blk" in r["Kernel_Name"]:
            cnt[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
rows = [(dur[k], cnt[k]) for k in dur if k in cnt and dur[k] > 1e-3]
rows = rows[len(rows) // 2:]          # second half of the run: clocks settled
if not rows:
    print(lib, "no data"); sys.exit()
n = len(rows)
ms = sum(r[0] for r in rows) / n * 1e3
g = sum(r[1].get("GRBM_GUI_ACTIVE", 0) for r in rows) / n / 8
avg = lambda k: sum(r[1].get(k, 0) for r in rows) / n
print("%-45s %2d launches: %.3f ms | %.2f M shader cycles | %.3f GHz | wave-cycles %.3f G (wait_any %.3f, wait_inst %.3f, active %.3f)" % (
    lib, n, ms, g / 1e6, g / (ms * 1e-3) / 1e9, avg("SQ_WAVE_CYCLES") / 1e9, avg("SQ_WAIT_ANY") / avg("SQ_WAVE_CYCLES"),
    avg("SQ_WAIT_INST_ANY") / avg("SQ_WAVE_CYCLES"), avg("SQ_ACTIVE_INST_ANY") / avg("SQ_WAVE_CYCLES")))
PY
  rm -rf $d
done
if [ "${SKIP_SMI:-0}" = "1" ]; then exit 0; fi
# un-profiled sustained loop of the first library with rocm-smi sampling
(HAVATAR_LIB=$PWD/$1 LAUNCHES=1500 python tools/march_once.py > /dev/null 2>&1 &) ; sleep 6
for i in 1 2 3; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr '\n' ' '; echo; sleep 1; done
wait
