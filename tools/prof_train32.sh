R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_train32
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/tools/train_bench.py --batch 32 --steps 40 --warmup 5 > $OUT/line.json 2> $OUT/log.txt
F=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
head -30 "$F" | cut -d, -f1-4 | cut -c1-170
cat $OUT/line.json | tail -1 | cut -c1-300
