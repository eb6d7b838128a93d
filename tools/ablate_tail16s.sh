#!/bin/bash
# A/B timing of tail16s with parts switched off (DIAG build only; numbers are diagnostic, never a result):
#   bash tools/ablate_tail16s.sh <outdir>     inside gpurun, after `make -C sr-for-cfd_amd/csrc DIAG=1`
OUT=${1:-gpurun_out/ablate_tail16s}; mkdir -p $OUT
R=${GRAFT_REPO_ROOT:-$(pwd)}
for A in ${ABLS:-0 1 2 4 8 16 32 7 56 63 128 256 191 319}; do
  SRCFD_BENCH_ALLOW_DIAG=1 SRCFD_LIB=$R/sr-for-cfd_amd/lib/libsrcfd_diag.so SRCFD_TAIL_ABLATE=$A timeout -k 10 100 python3 $R/bench.py --no-extras --no-cpu-baseline --steps 30 > $OUT/a$A.json 2> $OUT/a$A.err
  python3 - <<PY
import json
try:
    d = json.loads(open("$OUT/a$A.json").read().strip().splitlines()[-1]); print("ablate $A: tail %.4f ms  step %.4f ms" % (d["kernels_ms"]["tail(convT2-4+out)"], d["ms_per_step"]))
except Exception as e: print("ablate $A failed", e)
PY
done
