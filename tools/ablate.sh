#!/bin/bash
# needs the diagnostic build: make -C sr-for-cfd_amd/csrc clean && make -C sr-for-cfd_amd/csrc -j8 DIAG=1 (the default build has no work-skipping switches)
# Tail-kernel ablation (diagnostic): times bench.py with parts of the tail switched off.
for A in 0 1 2 4 8 6 14 15; do
  SRCFD_BENCH_ALLOW_DIAG=1 SRCFD_TAIL_ABLATE=$A python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ablate=$A tail ms', d['kernels_ms']['tail(convT2-4+out)'])"
done
