#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel trace + PMC passes) into the text summary committed under profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
print("# rocprofv3 summary for", out)
try:
    print("# bench line:", open(os.path.join(out, "bench_line.json")).read().strip()[:2000])
except OSError:
    pass
for f in glob.glob(os.path.join(out, "kt", "**", "*kernel_stats.csv"), recursive=True):
    print("\n## kernel stats (", os.path.relpath(f, out), ")")
    rows = list(csv.DictReader(open(f)))
    for r in rows[:12]:
        print("  %-70s calls=%-6s total_ns=%-14s avg_ns=%-12s pct=%s" % (r.get("Name", "")[:70], r.get("Calls"), r.get("TotalDurationNs"),
                                                                      r.get("AverageNs"), r.get("Percentage")))
for f in glob.glob(os.path.join(out, "kt", "**", "*kernel_trace.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    byk = defaultdict(list)
    for r in rows:
        byk[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r))
    print("\n## kernel trace: per-kernel durations and resources")
    for k, v in sorted(byk.items(), key=lambda kv: -sum(d for d, _ in kv[1]))[:8]:
        d = sorted(x for x, _ in v)
        r = v[0][1]
        print("  %-60s n=%-4d med_us=%-10.1f min_us=%-10.1f VGPR=%s AGPR=%s SGPR=%s LDS=%s scratch=%s grid=%s wg=%s" % (
            k[:60], len(d), d[len(d) // 2] / 1e3, d[0] / 1e3, r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"),
            r.get("LDS_Block_Size"), r.get("Scratch_Size"), r.get("Grid_Size"), r.get("Workgroup_Size")))
import json
pmc_json = {}
print("\n## PMC (per-dispatch mean over the dispatches of each kernel)")
for f in sorted(glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    rows = list(csv.DictReader(open(f)))
    acc = defaultdict(lambda: defaultdict(list))
    for r in rows:
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        if not any(t in k for t in ("march", "fba", "ufd", "plane_project", "conv3x3", "gemm_split", "upconv", "haar", "style_demod", "absmax")):
            continue
        print("  %s" % k[:80])
        for c, vals in sorted(cs.items()):
            print("      %-32s mean=%.6g  (n=%d)" % (c, sum(vals) / len(vals), len(vals)))
            pmc_json.setdefault(k.split("(")[0].replace("void ", "").strip(), {})[c] = sum(vals) / len(vals)
# calibration: counters of a streaming launch of known size
try:
    cal = json.load(open(os.path.join(out, "calib.json")))
    got = {}
    for f in sorted(glob.glob(os.path.join(out, "calib_*", "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            if "fba_vec_kernel" in r["Kernel_Name"]:
                got.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    cal["FETCH_SIZE_KB"] = sum(got["FETCH_SIZE"]) / len(got["FETCH_SIZE"])
    cal["WRITE_SIZE_KB"] = sum(got["WRITE_SIZE"]) / len(got["WRITE_SIZE"])
    cal["read_bytes_per_FETCH_KB"] = cal["read_bytes"] / cal["FETCH_SIZE_KB"]
    cal["write_bytes_per_WRITE_KB"] = cal["write_bytes"] / cal["WRITE_SIZE_KB"]
    pmc_json["calibration"] = cal
    print("\n## counter calibration (fused_bias_act [4,64,512,512] f32: %d B read, %d B written per launch)" % (cal["read_bytes"], cal["write_bytes"]))
    print("   FETCH_SIZE=%.1f KB -> %.1f B per reported KB;  WRITE_SIZE=%.1f KB -> %.1f B per reported KB" % (
        cal["FETCH_SIZE_KB"], cal["read_bytes_per_FETCH_KB"], cal["WRITE_SIZE_KB"], cal["write_bytes_per_WRITE_KB"]))
except Exception as e:  # noqa
    print("\n## counter calibration: unavailable (%r)" % (e,))
json.dump(pmc_json, open(os.path.join(out, "pmc.json"), "w"), indent=1, sort_keys=True)
