#!/bin/bash
# Per-kernel timeline of ONE steady-state frame (eager launches, after MIOpen's solver search): which kernels make up the
# encoder phase and how long each takes.  Run on the GPU box: gpurun -- 'bash tools/frame_timeline.sh'
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/timeline
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- python bench.py --steps 4 --warmup 4 --graph ${GRAPH:-0} --no-cpu-baseline --extras 0 --live-pmc 0 > $OUT/bench.log 2>&1
python - <<'PY' > $OUT/frame_timeline.txt
import csv, glob, collections
f = glob.glob("gpurun_out/timeline/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# (in fp16 mode every frame also dispatches the range guard's bf16 stand-in, which returns at once: only the real march counts)
march = [i for i, r in enumerate(rows) if "hav_march" in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 1000000]
# the timed steps are the last ones: take the dispatches between the 2nd-last and the last march launch of the timed loop
timed = march[4:8] if len(march) >= 8 else march[-2:]
a, b = timed[-2], timed[-1]
frame = rows[a + 1:b + 1]
t0 = int(frame[0]["Start_Timestamp"]); t1 = int(frame[-1]["End_Timestamp"])
print("# one frame: %d kernel launches, %.3f ms from first start to last end" % (len(frame), (t1 - t0) / 1e6))
agg = collections.OrderedDict()
busy = 0
for r in frame:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); busy += d
    k = r["Kernel_Name"][:90]
    e = agg.setdefault(k, [0, 0]); e[0] += 1; e[1] += d
print("# sum of kernel durations %.3f ms (gaps %.3f ms)" % (busy / 1e6, (t1 - t0 - busy) / 1e6))
for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%8.1f us  n=%-4d avg %7.1f us  %s" % (d / 1e3, n, d / 1e3 / n, k))
print("\n# in launch order")
for r in frame:
    print("%8.1f us  %s  grid=%s wg=%s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:80], r.get("Grid_Size"), r.get("Workgroup_Size")))
PY
rm -rf $OUT/kt
head -45 $OUT/frame_timeline.txt
