#!/bin/bash
# PMC pass over the DIAG build with one wave class switched off (diagnostic): bash tools/pmc_tail_x.sh <outdir> <ablate> "<counters>"
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/$1; A=$2; CTRS=$3; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SRCFD_BENCH_ALLOW_DIAG=1 SRCFD_LIB=$R/sr-for-cfd_amd/lib/libsrcfd_diag.so SRCFD_TAIL_ABLATE=$A
timeout -k 10 150 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/*/*_counter_collection.csv"):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if 'tail16' not in r['Kernel_Name']: continue
        agg[r['Kernel_Name'].split('(')[0][-30:]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()): print("ablate $A", k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
