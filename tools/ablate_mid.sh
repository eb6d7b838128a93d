#!/bin/bash
for A in 0 1 2 3 4 7; do
  SRCFD_MID_ABLATE=$A python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('mid ablate=$A', d['kernels_ms']['mid(convT0+convT1)'])"
done
