#!/bin/bash
# Steady-state kernel breakdown of one training step (tools/bench_train.py): kernels of the LAST backward+forward window.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/train_tl
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- env BENCH_TRAIN_NO_BREAKDOWN=1 python tools/bench_train.py > $OUT/log.txt 2>&1
python - <<'PY' > $OUT/train_timeline.txt
import csv, glob, collections
f = glob.glob("gpurun_out/train_tl/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the first field-input kernel of a step marks its phase (4 per step: 2 ray chunks x 2 passes); with HAVATAR_HIP_TRAIN=0 fall back
# to the optimiser's multi-tensor kernels.  The window runs from that kernel of one step to the same kernel of the next one.
fi = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("void field_inputs_kernel<0>")]
if len(fi) >= 12:
    a, b = fi[-8] - 1, fi[-4] - 1
else:
    adam = [i for i, r in enumerate(rows) if "multi_tensor_apply" in r["Kernel_Name"] or "adam" in r["Kernel_Name"].lower()]
    ends = [adam[i] for i in range(len(adam)) if i + 1 == len(adam) or adam[i + 1] - adam[i] > 50]
    a, b = ends[-3], ends[-2]
frame = rows[a + 1:b + 1]
t0, t1 = int(frame[0]["Start_Timestamp"]), int(frame[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in frame)
print("# one training step: %d launches, span %.1f ms, kernel time %.1f ms" % (len(frame), (t1 - t0) / 1e6, busy / 1e6))
agg = collections.OrderedDict()
for r in frame:
    e = agg.setdefault(r["Kernel_Name"][:100] if "at::native" not in r["Kernel_Name"] else r["Kernel_Name"].replace("at::native::", "").replace("(anonymous namespace)::", "")[:330], [0, 0]); e[0] += 1; e[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    print("%9.2f ms  n=%-5d avg %9.1f us  %s" % (d / 1e6, n, d / 1e3 / n, k))
PY
rm -rf $OUT/kt
cat $OUT/train_timeline.txt
