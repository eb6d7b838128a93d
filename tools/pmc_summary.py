#!/usr/bin/env python3
"""Per-kernel, per-launch averages of rocprofv3 --pmc passes.

    python tools/pmc_summary.py gpurun_out/prof_<tag> > profiles/<round>/<name>_pmc_per_launch_avg.json
Reads every *counter_collection.csv below the directory (one pass per counter group, as tools/profile_run.sh
writes them), sums a counter over its dimensions (XCC / SE instances) within one dispatch, and averages over
the dispatches of each kernel."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
per = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))  # kernel -> counter -> dispatch -> value
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"].split("(")[0]
            per[k][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
out = {k: {c: sum(d.values()) / len(d) for c, d in sorted(cs.items())} for k, cs in sorted(per.items())}
json.dump(out, sys.stdout, indent=1)
