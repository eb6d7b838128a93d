#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (inside gpurun):
#   1. --kernel-trace --stats of the DEFAULT command (python3 bench.py: headline + parity_path + train + tiled legs)
#   2. separate --pmc passes (never combined with other trace domains) over the bf16 headline and over the f32 parity path:
#      HBM bytes (FETCH_SIZE, WRITE_SIZE), LDS conflicts, VALU / MFMA instruction counts and busy cycles, wave wait cycles
# Usage: bash tools/profile_run.sh <tag>        -> gpurun_out/prof_<tag>/...; summarise with tools/pmc_summary.py
set -o pipefail
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --no-cpu-baseline > $OUT/stats_bench_line.json 2> $OUT/stats.log
for PREC in bf16 fp32; do
  BENCH="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --precision $PREC"
  for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
    N=$(echo $C | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_${PREC}/$N -- $BENCH > $OUT/pmc_${PREC}_$N.log 2>&1 || echo "pmc pass $PREC $N failed" >> $OUT/errors.txt
  done
done
find $OUT -name "*.csv" | head -80 > $OUT/files.txt
