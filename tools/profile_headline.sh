#!/bin/bash
# Headline-only rocprofv3 kernel stats: the roofline's kernel duration reproducible from profiles/.
#   bash tools/profile_headline.sh <tag>   (inside gpurun) -> gpurun_out/prof_<tag>/f_headline_only_{bf16,fp32}_*
# One precision per run, --no-extras (no parity / train / tiled / host legs), 300 timed steps: every launch of the
# dominant kernel in the trace is a 768-sample launch.
set -o pipefail
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for PREC in bf16 fp32; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/headline_$PREC -- python3 $R/bench.py --no-cpu-baseline --no-extras --precision $PREC --steps 300 --warmup 30 \
      > $OUT/f_headline_only_${PREC}_bench_line.json 2> $OUT/headline_$PREC.log || echo "headline $PREC failed" >> $OUT/errors.txt
  F=$(find $OUT/headline_$PREC -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && cp "$F" $OUT/f_headline_only_${PREC}_kernel_stats.csv
  T=$(find $OUT/headline_$PREC -name "*kernel_trace.csv" | head -1)
  [ -n "$T" ] && python3 $R/tools/trace_gaps.py "$T" > $OUT/f_headline_only_${PREC}_trace_gaps.txt 2>&1
  rm -rf $OUT/headline_$PREC
done
