#!/bin/bash
# Memory-side counters of the streaming ops at the bench_ops.py shapes (each launch on its own buffers): HBM bytes fetched / written per
# launch, L1 -> L2 read requests, texture-addresser busy cycles.  Run on the GPU box: gpurun -- 'bash tools/pmc_ops.sh'
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_ops
rm -rf $OUT; mkdir -p $OUT
for pass in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  n=$(echo $pass | tr ' ' '_' | cut -c1-30)
  timeout 400 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/$n -o pmc -- python tools/bench_ops.py > $OUT/$n.log 2>&1
done
python - <<'PY' > $OUT/pmc_ops.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_ops/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        if not any(s in k for s in ("ufd_", "fba_vec", "haar")): continue
        acc[k + " grid=" + r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c, v in sorted(acc[k].items()):
        print("    %-34s mean %14.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
find $OUT -mindepth 1 -maxdepth 1 -type d -exec rm -r {} +
rm -f $OUT/*.log
cat $OUT/pmc_ops.txt
