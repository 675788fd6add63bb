#!/bin/bash
# Counter passes over tools/march_once.py (the production ray-march kernel on one 512^2 frame) for ONE library build:
#   gpurun -- 'bash tools/pmc_march.sh [lib.so] [tag]'   -> gpurun_out/pmc_<tag>.txt  (per-launch means of the march kernel)
# Counters are collected in their own passes with --kernel-trace only (MI355X_MICROARCH.md, rocprofv3 section).
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
LIB=${1:-}; TAG=${2:-run}
[ -n "$LIB" ] && export HAVATAR_LIB=$PWD/$LIB
OUT=gpurun_out/pmc_$TAG; rm -rf $OUT; mkdir -p $OUT
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" \
            "SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE" \
            "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  n=$(echo $pass | tr ' ' '_' | cut -c1-40)
  LAUNCHES=3 timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/p_$n -o pmc -- python tools/march_once.py > $OUT/p_$n.log 2>&1
done
python - "$OUT" <<'PY' | tee gpurun_out/pmc_$TAG.txt
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(list)
for f in glob.glob(os.path.join(out, "p_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "hav_march_blk_kernel<1, 2, 2>" in r["Kernel_Name"] or "hav_march_blk_kernel<0, 2, 2>" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
for k in sorted(m): print("%-34s %16.0f" % (k, m[k]))
if "GRBM_GUI_ACTIVE" in m:
    cyc = m["GRBM_GUI_ACTIVE"] / 8
    print("cycles per launch (GRBM/8 XCDs) %.0f" % cyc)
    for nm, key, div in (("TA busy", "TA_TA_BUSY_sum", 256), ("MFMA busy", "SQ_VALU_MFMA_BUSY_CYCLES", 1024), ("VALU busy (x4 quad-cycles)", "SQ_ACTIVE_INST_VALU", 256), ("LDS idx active", "SQ_LDS_IDX_ACTIVE", 256), ("LDS bank conflict", "SQ_LDS_BANK_CONFLICT", 256)):
        if key in m: print("  %-28s %.3f" % (nm, m[key] / div / cyc))
PY
rm -rf $OUT
