export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out/prof_graph
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace -d $OUT -o bench -- python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $OUT/stdout.log 2> $OUT/stderr.log
DB=$(find $OUT -name "*.db" | head -1)
python $REPO/tools/rocpd_gaps.py $DB 5
python $REPO/tools/rocpd_gaps.py $DB 5 head > $OUT/head.txt
python $REPO/tools/rocpd_summary.py $DB 13 > $REPO/gpurun_out/graph_mode_stats.md
rm -f $DB
