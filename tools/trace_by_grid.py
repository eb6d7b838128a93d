#!/usr/bin/env python3
"""Groups a rocprofv3 kernel_trace.csv by (kernel, grid) -> count, avg us, total ms."""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: [0, 0.0])
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"].split("(")[0][-48:]
        key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
        a = acc[key]
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(a[1] for a in acc.values())
for k, a in sorted(acc.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{k[0]:48s} grid=({k[1]},{k[2]},{k[3]}) n={a[0]:4d} avg={a[1] / a[0]:9.1f}us total={a[1] / 1e3:8.2f}ms {100 * a[1] / tot:5.1f}%")
