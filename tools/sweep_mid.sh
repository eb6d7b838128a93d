#!/bin/bash
# A/B of the mid kernel's workgroup size (diagnostic)
for W in 4 8 16; do
  SRCFD_MID_WAVES=$W python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('mid waves=$W', d['kernels_ms']['mid(convT0+convT1)'], 'total', d['ms_per_step'])"
done
