#!/bin/bash
# Kernel durations (rocprofv3 --kernel-trace --stats, no events in the stream) and HBM fetch bytes of the bf16 headline under two values
# of one library switch:  bash tools/ab_prof.sh SRCFD_MID 1 2   -> gpurun_out/ab_<name>/<value>/...
set -o pipefail
NAME=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/ab_$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SRCFD_BENCH_ALLOW_DIAG=1
for V in "$@"; do
  export $NAME=$V; mkdir -p $OUT/$V
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$V/stats -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 10 > $OUT/$V/stats_line.json 2> $OUT/$V/stats.log
  for C in FETCH_SIZE WRITE_SIZE "${EXTRA_PMC:-GRBM_GUI_ACTIVE}"; do
    N=$(echo $C | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$V/pmc/$N -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 > $OUT/$V/pmc_$N.log 2>&1
  done
  python3 $R/tools/pmc_summary.py $OUT/$V/pmc > $OUT/$V/pmc.json 2>/dev/null
  F=$(find $OUT/$V/stats -name "*kernel_stats.csv" | head -1)
  echo "== $NAME=$V"; grep -E "tail16|mid16|enc16|dense1_16" "$F" | cut -d, -f1-4,6,7
  python3 - <<PY
import json
d=json.load(open("$OUT/$V/pmc.json"))
for k,v in d.items():
    if "tail16" in k or "mid16" in k: print(k[:40], "FETCH x2 MB", round(v.get("FETCH_SIZE",0)*2048/1e6,1), "WRITE MB", round(v.get("WRITE_SIZE",0)*1024/1e6,1), {c: round(x) for c, x in v.items() if c not in ("FETCH_SIZE", "WRITE_SIZE")})
PY
done
