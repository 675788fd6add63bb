#!/bin/bash
# rocprofv3 recipe (run on the GPU box through gpurun): kernel trace + stats, then separate PMC passes.
# usage: tools/profile.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/...
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=${1:-run}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
BENCH="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --live-pmc 0 --extras 0 $*"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $BENCH > $OUT/bench_kt.log 2>&1
grep -m1 "^{\"metric\"" $OUT/bench_kt.log > $OUT/bench_line.json
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" \
            "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
            "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  n=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$n -o pmc -- $BENCH > $OUT/pmc_$n.log 2>&1
done
# calibration of the memory-side counters on a launch of known size (MI355X_MICROARCH.md, HBM section)
for pass in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/calib_$pass -o pmc -- python tools/bench_ops.py --calib > $OUT/calib_$pass.log 2>&1
done
grep -h -m1 calib_kernel $OUT/calib_FETCH_SIZE.log > $OUT/calib.json
python tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cp $OUT/kt/kt_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
# keep only the compact artefacts (gpurun copies back <= 64 MiB)
find $OUT -mindepth 1 -maxdepth 1 -type d -exec rm -r {} +
rm -f $OUT/*.log
cat $OUT/summary.txt
