set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
CASE=${1:-blur}
OUT=gpurun_out/ufd_pmc_$CASE; rm -rf $OUT; mkdir -p $OUT
i=0
for pass in "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU" \
            "SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
            "TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
            "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
            "TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_32B_sum" \
            "FETCH_SIZE WRITE_SIZE" \
            "TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum" \
            "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_TAG_STALL_sum TCC_BUSY_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/p$i -o pmc -- python tools/ufd_pmc_drv.py $CASE > $OUT/p$i.log 2>&1 || tail -3 $OUT/p$i.log
done
python - "$OUT" <<'PY' | tee gpurun_out/ufd_pmc_$CASE.txt
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not any(s in k for s in ("ufd_", "fba_vec")): continue
        acc[k[:60] + " grid=" + r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in acc for c in acc[k]})
keys = sorted(acc)
for k in keys: print("#", k)
for c in names:
    print("%-34s" % c + "".join("%16.0f" % (sum(acc[k][c]) / max(1, len(acc[k][c]))) for k in keys))
PY
find $OUT -mindepth 1 -maxdepth 1 -type d -exec rm -r {} +
