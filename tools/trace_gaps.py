#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 kernel_trace.csv, by (previous kernel -> next kernel) pair.
Usage: trace_gaps.py kernel_trace.csv [last_n_kernels]   (default: the last 200 launches, i.e. the timed steps)"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-(int(sys.argv[2]) if len(sys.argv) > 2 else 200):]
short = lambda n: n.split("(")[0].replace("void srcfd::", "")[:34]
gaps, durs = defaultdict(list), defaultdict(list)
for a, b in zip(rows, rows[1:]):
    gaps[(short(a["Kernel_Name"]), short(b["Kernel_Name"]))].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
for r in rows:
    durs[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("kernel durations (us): avg / min / n")
for k, v in durs.items():
    print(f"  {k:36s} {sum(v) / len(v) / 1e3:8.2f} {min(v) / 1e3:8.2f} {len(v):5d}")
print("idle between kernels (us): avg / min / max / n")
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {k[0]:36s} -> {k[1]:36s} {sum(v) / len(v) / 1e3:7.2f} {min(v) / 1e3:7.2f} {max(v) / 1e3:7.2f} {len(v):5d}")
