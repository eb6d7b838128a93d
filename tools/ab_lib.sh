#!/bin/bash
# Two builds of the library against each other on one box: alternating bench runs (the library is loaded once per process, so this
# cannot be an in-process A/B like tools/ab_switch.py).  usage: bash tools/ab_lib.sh <other libsrcfd.so> [rounds]
OTHER=$1; N=${2:-3}
export SRCFD_BENCH_ALLOW_DIAG=1
for i in $(seq $N); do
  for L in "" "$OTHER"; do
    SRCFD_LIB=$L python3 bench.py --no-extras --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${L:-default}', d['ms_per_step'], d['kernels_ms'])"
  done
done
