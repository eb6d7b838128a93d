#!/usr/bin/env python3
"""The launches of the last `n` kernels of a rocprofv3 kernel_trace.csv in start order: start offset (us), duration (us), queue, name.
Usage: trace_last_step.py kernel_trace.csv [n]   (n = kernels per step; default 43 = one training step)"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-(int(sys.argv[2]) if len(sys.argv) > 2 else 43):]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} q{r.get('Queue_Id', '?'):>2} {r['Kernel_Name'].split('(')[0].replace('void srcfd::', '')[:60]} grid={r.get('Grid_Size_X', r.get('Grid_Size', ''))}")
