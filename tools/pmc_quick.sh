#!/bin/bash
# Quick PMC passes for the tail kernels (inside gpurun): bash tools/pmc_quick.sh <tag> [bench args, e.g. --precision fp32]
TAG=${1:-q}; shift; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras $*"
timeout -k 10 150 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/a -- $BENCH > $OUT/a.log 2>&1
timeout -k 10 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/b -- $BENCH > $OUT/b.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/*/")):
    for f in glob.glob(d+"/*/*_counter_collection.csv"):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)): agg[r["Kernel_Name"].split("(")[0][-40:]+" g"+r["Grid_Size"]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in agg.items():
            if 'srcfd' in k or 'tail' in k: print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
