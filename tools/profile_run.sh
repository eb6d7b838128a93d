#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box:
#   1. --kernel-trace --stats            (per-kernel time)
#   2. separate --pmc passes             (HBM bytes, LDS conflicts, VALU/MFMA busy)
# Usage (inside gpurun): bash tools/profile_run.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats.log 2>&1
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -- $BENCH > $OUT/pmc_$N.log 2>&1 || echo "pmc pass $N failed" >> $OUT/errors.txt
done
find $OUT -name "*.csv" | head -50 > $OUT/files.txt
