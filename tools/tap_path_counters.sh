#!/bin/bash
# Texture-path counters of the march kernel (TA busy, L1 tag requests, L1 -> L2 read requests, stalls) for the in-tree library and,
# if present, for an alternative build (HAVATAR_LIB), e.g. the flat plane layout -DHAV_TG=0.  Run on the GPU box through gpurun:
#   gpurun -- 'bash tools/tap_path_counters.sh [alt.so]'  -> gpurun_out/tap_path_counters.txt
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/tapc; rm -rf $OUT; mkdir -p $OUT
BENCH="python bench.py --steps 4 --warmup 1 --no-cpu-baseline"
run() {   # $1 = tag, env HAVATAR_LIB optional
  i=0
  for pass in "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
              "TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/$1_$i -o pmc -- $BENCH > $OUT/$1_$i.log 2>&1
  done
}
run tree
if [ -n "${1:-}" ]; then HAVATAR_LIB=$PWD/$1 run alt; fi
python - <<'PY' > gpurun_out/tap_path_counters.txt
import csv, glob, collections
for tag in ("tree", "alt"):
    acc = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/tapc/%s_*/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            if "hav_march" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if acc:
        print("## %s (per launch of the march kernel, mean over launches)" % ("in-tree library" if tag == "tree" else "alternative library"))
        for k, v in sorted(acc.items()):
            print("   %-40s %.6g" % (k, sum(v) / len(v)))
PY
rm -rf $OUT
cat gpurun_out/tap_path_counters.txt
