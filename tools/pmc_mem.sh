#!/bin/bash
# HBM traffic pass (inside gpurun): bash tools/pmc_mem.sh <tag> [bench args] -- FETCH_SIZE (KB; x2 on gfx950 for wide reads) / WRITE_SIZE per launch
TAG=${1:-m}; shift; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/pmcmem_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras $*"
# one counter per pass: FETCH_SIZE and WRITE_SIZE together exceed what one pass can collect on gfx950
timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/a -- $BENCH > $OUT/a.log 2>&1
timeout -k 10 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/b -- $BENCH > $OUT/b.log 2>&1
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/*/*/*_counter_collection.csv"):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)): agg[r["Kernel_Name"].split("(")[0][-40:]+" g"+r["Grid_Size"]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        if 'srcfd' in k: print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
