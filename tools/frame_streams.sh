#!/bin/bash
# The frame's hipGraph as the hardware runs it: per queue (= stream of the captured graph) the kernels of ONE replay with start offsets,
# so that the critical path of the encoder phase (two generators on two streams, then preparation + march) can be read off.
# Run on the GPU box: gpurun -- 'bash tools/frame_streams.sh'
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/frame_streams
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- python bench.py --steps 4 --warmup 4 --graph 1 --no-cpu-baseline --extras 0 --live-pmc 0 > $OUT/bench.log 2>&1
python - <<'PY' > $OUT/frame_streams.txt
import csv, glob, collections
f = glob.glob("gpurun_out/frame_streams/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
march = [i for i, r in enumerate(rows) if "hav_march_blk_kernel<1, 3, 2>" in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 1000000]
a, b = march[-2], march[-1]
frame = rows[a + 1:b + 1]
t0 = int(rows[a]["End_Timestamp"])
print("# one graph replay: %d kernels between the end of one march and the end of the next; times in us after the previous march's end" % len(frame))
qkey = "Queue_Id" if "Queue_Id" in frame[0] else ("Stream_Id" if "Stream_Id" in frame[0] else None)
print("# columns of the trace:", ",".join(frame[0].keys()))
byq = collections.OrderedDict()
for r in frame:
    byq.setdefault(r.get(qkey, "?") if qkey else "?", []).append(r)
for q, rs in byq.items():
    s0, e1 = int(rs[0]["Start_Timestamp"]) - t0, int(rs[-1]["End_Timestamp"]) - t0
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    print("\n## queue %s: %d kernels, first start %.1f, last end %.1f, busy %.1f us" % (q, len(rs), s0 / 1e3, e1 / 1e3, busy / 1e3))
    prev_end = None
    for r in rs:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        print("  %9.1f  +%6.1f gap  %7.1f us  %s" % (s / 1e3, gap, (e - s) / 1e3, r["Kernel_Name"][:70]))
        prev_end = e
PY
rm -rf $OUT/kt
head -30 $OUT/frame_streams.txt
