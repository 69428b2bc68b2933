export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out/prof_viz
mkdir -p $OUT
cd /tmp
D2P_GRAPH=0 rocprofv3 --kernel-trace --stats -d $OUT -o bench -- python $REPO/bench.py --preset vizdoom --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/stdout.log 2> $OUT/stderr.log
DB=$(find $OUT -name "*.db" | head -1)
python $REPO/tools/rocpd_summary.py $DB 8 > $REPO/gpurun_out/r01q_kernel_stats_vizdoom.md
rm -f $DB
