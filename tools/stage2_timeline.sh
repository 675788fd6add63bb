#!/bin/bash
# Per-kernel timeline of ONE steady-state SWGAN_unet forward (512 -> 1024, BASELINE config 4; hipGraph replays of tools/bench_stage2.py):
# the kernels between the last two final-synthesis launches.  Run on the GPU box: gpurun -- 'bash tools/stage2_timeline.sh'
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/s2_timeline
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- python tools/bench_stage2.py > $OUT/bench.log 2>&1
python - <<'PY' > $OUT/stage2_timeline.txt
import csv, glob, collections
f = glob.glob("gpurun_out/s2_timeline/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
last = rows[-1]["Kernel_Name"]
ends = [i for i, r in enumerate(rows) if r["Kernel_Name"] == last]
# the final synthesis is the forward's last launch; earlier synthesis launches of the same forward share the name, so step back
# by the per-forward count (found from the spacing of the last dispatches)
gaps = [ends[i + 1] - ends[i] for i in range(len(ends) - 1)]
# the launch pattern repeats with the forward: find the period of the gap sequence (e.g. 6 synthesis launches per forward)
tail = gaps[-48:]
period = next(p for p in range(1, 25) if all(tail[i] == tail[i - p] for i in range(p, len(tail))))
per = sum(tail[-period:])
a = len(rows) - 1 - per
frame = rows[a + 1:]
t0 = int(frame[0]["Start_Timestamp"]); t1 = int(frame[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in frame)
print("# one SWGAN_unet forward (512 -> 1024): %d kernel launches, %.3f ms from first start to last end, sum of kernel durations %.3f ms" % (len(frame), (t1 - t0) / 1e6, busy / 1e6))
agg = collections.OrderedDict()
for r in frame:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    e = agg.setdefault(r["Kernel_Name"][:100], [0, 0]); e[0] += 1; e[1] += d
for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%8.1f us  n=%-4d avg %7.1f us  %s" % (d / 1e3, n, d / 1e3 / n, k))
PY
rm -rf $OUT/kt
head -40 $OUT/stage2_timeline.txt
