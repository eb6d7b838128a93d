#!/bin/bash
# tail32 ablation (diagnostic): times the f32-grade tail with parts switched off (Tail32Params::ablate).
# needs the diagnostic build: make -C sr-for-cfd_amd/csrc -j8 DIAG=1   (the default build has no work-skipping switches)
#   bash tools/ablate_tail32.sh [precision=fp32x3] [list of masks]
R=$(cd "$(dirname "$0")/.." && pwd)
P=${1:-fp32x3}; shift
for A in ${@:-0 1 2 4 8 16 32 64 66 3 7 127}; do
  SRCFD_BENCH_ALLOW_DIAG=1 SRCFD_LIB=$R/sr-for-cfd_amd/lib/libsrcfd_diag.so SRCFD_TAIL32_ABLATE=$A python3 $R/bench.py --precision $P --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('precision=$P ablate=$A tail32 ms', [v for n,v in k.items() if 'output_image' in n or 'convT2' in n or 'transpose_2' in n], 'step', d['ms_per_step'])"
done
