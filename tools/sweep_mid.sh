#!/bin/bash
# A/B of the mid kernel's workgroup shape (diagnostic): SRCFD_MID=1 (8 waves x 32 pixels), 2 (8 x 64), 3 (4 x 64, shipped), and the 32-pixel-per-wave
# shapes with 4 / 16 waves (SRCFD_MID_WAVES).  One process per value: for a comparison on one box and clock state use tools/ab_switch.py SRCFD_MID 1 2 3.
export SRCFD_BENCH_ALLOW_DIAG=1
for M in 1 2 3; do
  SRCFD_MID=$M python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('SRCFD_MID=$M', d['kernels_ms']['mid(convT0+convT1)'], 'total', d['ms_per_step'])"
done
for W in 4 16; do
  SRCFD_MID_WAVES=$W python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('SRCFD_MID_WAVES=$W', d['kernels_ms']['mid(convT0+convT1)'], 'total', d['ms_per_step'])"
done
