#!/usr/bin/env python3
"""How much of a rocprofv3 kernel_trace.csv ran concurrently: sum of kernel durations vs the union of their intervals,
per queue and overall.  Usage: trace_overlap.py kernel_trace.csv [skip_first_n]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(sys.argv[2]) if len(sys.argv) > 2 else 0:]
iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")) for r in rows]
tot = sum(e - s for s, e, _ in iv)
union, cur_s, cur_e = 0, None, None
for s, e, _ in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
perq = defaultdict(lambda: [0, 0])
for s, e, q in iv:
    perq[q][0] += 1
    perq[q][1] += e - s
print(f"kernels {len(iv)}  sum of durations {tot / 1e6:.3f} ms  union {union / 1e6:.3f} ms  span {(iv[-1][1] - iv[0][0]) / 1e6:.3f} ms")
for q, (n, t) in sorted(perq.items()):
    print(f"  queue {q}: {n} kernels, {t / 1e6:.3f} ms")
