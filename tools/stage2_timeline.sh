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
# the trace ends with the 10 timed hipGraph replays of the 512 -> 1024 forward: identical launch sequences, so the forward is the shortest
# period P of the kernel-name sequence at the tail (names[-P:] == names[-2P:-P] == names[-3P:-2P])
names = [r["Kernel_Name"] for r in rows]
per = next(p for p in range(20, 2000) if names[-p:] == names[-2 * p:-p] and names[-p:] == names[-3 * p:-2 * p])
frame = rows[-per:]
t0 = int(frame[0]["Start_Timestamp"]); t1 = int(frame[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in frame)
print("# one SWGAN_unet forward (512 -> 1024): %d kernel launches, %.3f ms from first start to last end, sum of kernel durations %.3f ms" % (len(frame), (t1 - t0) / 1e6, busy / 1e6))
agg = collections.OrderedDict()
for r in frame:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    e = agg.setdefault(r["Kernel_Name"][:100], [0, 0]); e[0] += 1; e[1] += d
for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%8.1f us  n=%-4d avg %7.1f us  %s" % (d / 1e3, n, d / 1e3 / n, k))
print("## in launch order: start after the forward's first launch | gap to the previous kernel's end | duration | grid x workgroup")
prev = t0
for r in frame:
    s0, e0 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f  +%6.1f gap %8.1f us  %-70s %sx%s" % ((s0 - t0) / 1e3, (s0 - prev) / 1e3, (e0 - s0) / 1e3, r["Kernel_Name"][:70], r.get("Grid_Size_X", "?"), r.get("Workgroup_Size_X", "?")))
    prev = e0
PY
rm -rf $OUT/kt
head -40 $OUT/stage2_timeline.txt
