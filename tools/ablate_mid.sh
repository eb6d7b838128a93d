#!/bin/bash
# needs the diagnostic build: make -C sr-for-cfd_amd/csrc clean && make -C sr-for-cfd_amd/csrc -j8 DIAG=1 (the default build has no work-skipping switches)
for A in ${ABLS:-0 1 2 3 4 7 8 16 24}; do
  SRCFD_BENCH_ALLOW_DIAG=1 SRCFD_MID_ABLATE=$A python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('mid ablate=$A', d['kernels_ms']['mid(convT0+convT1)'])"
done
