#!/bin/bash
# kernel-trace of a few training steps -> gpurun_out/prof_train/<tag>_kernels.txt   (inside gpurun)
TAG=${1:-train}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_train
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$TAG -o $TAG -- python3 $R/tools/train_bench.py --steps 10 "$@" > $OUT/$TAG.log 2>&1
F=$(find $OUT/$TAG -name "*kernel_stats.csv" | sort | tail -1)
if [ -n "$F" ]; then cp $F $OUT/${TAG}_kernel_stats.csv; fi
T=$(find $OUT/$TAG -name "*kernel_trace.csv" | sort | tail -1)
if [ -n "$T" ]; then python3 $R/tools/trace_by_grid.py $T > $OUT/${TAG}_by_grid.txt; python3 $R/tools/trace_overlap.py $T 300 > $OUT/${TAG}_overlap.txt; rm -f $T; fi
