#!/bin/bash
# One rocprofv3 PMC pass over bench.py (inside gpurun): bash tools/pmc_pass.sh <tag> "<counters>" [bench args]
TAG=$1; CTRS=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 150 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/*/*_counter_collection.csv"):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if 'srcfd' not in r['Kernel_Name']: continue
        k=r['Kernel_Name'].split('(')[0].replace('void srcfd::','')[:28]+' g'+r['Grid_Size']
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()): print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
