#!/bin/bash
# mid16 section timers (make DIAG=1 build) and work-skipping ablations for SRCFD_MID=1 (256-pixel workgroups) and =2 (512-pixel)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export SRCFD_LIB=$R/sr-for-cfd_amd/lib/libsrcfd_diag.so SRCFD_BENCH_ALLOW_DIAG=1
for M in ${MIDS:-1 2}; do
  echo "== SRCFD_MID=$M section timers"
  SRCFD_MID=$M SRCFD_MID_PROF=1 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep -A14 "^mid16" | cut -c1-220
  for A in ${ABLS:-0 1 2 4 7 16}; do
    SRCFD_MID=$M SRCFD_MID_ABLATE=$A python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('SRCFD_MID=$M ablate=$A', d['kernels_ms']['mid(convT0+convT1)'])"
  done
done
