#!/bin/bash
# The launches of the LAST training step of a short run, in start order with stream ids, gaps and durations (inside gpurun):
#   bash tools/prof_train_step.sh <batch> [n_launches]  -> gpurun_out/prof_train/step_b<batch>.txt
B=${1:-8}; N=${2:-60}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_train
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/step_b$B -o t -- python3 $R/tools/train_bench.py --batch $B --steps 12 > $OUT/step_b$B.log 2>&1
T=$(find $OUT/step_b$B -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_last_step.py "$T" $N > $OUT/step_b$B.txt
rm -rf $OUT/step_b$B
