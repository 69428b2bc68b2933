#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel table of BASELINE config 4 (ViZDoom 80x80x3, k=10, B=32), eager
# one-stream launches (per-kernel durations), + the HBM traffic counters of the same command in separate passes.
# usage: bash tools/profile_vizdoom.sh <tag> [pmc]      (outputs gpurun_out/<tag>_kernel_stats_vizdoom.md, ..._pmc_traffic_vizdoom.*)
TAG=${1:-r05}
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out/prof_viz_$TAG
mkdir -p $OUT
cd /tmp
export D2P_GRAPH=0 D2P_SIDE_STREAM=0
ARGS="--preset vizdoom --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-h2d"
rocprofv3 --kernel-trace --stats -d $OUT -o bench -- python $REPO/bench.py $ARGS > $OUT/stdout.log 2> $OUT/stderr.log
DB=$(find $OUT -name "*.db" | head -1)
python $REPO/tools/rocpd_summary.py $DB 8 > $REPO/gpurun_out/${TAG}_kernel_stats_vizdoom.md
rm -f $DB
if [ "$2" = "pmc" ]; then
  PM=$REPO/gpurun_out/pmc_viz_$TAG
  mkdir -p $PM
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C -d $PM -o pmc_$C -- python $REPO/bench.py --preset vizdoom --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-h2d > $PM/${C}_stdout.log 2> $PM/${C}_stderr.log
  done
  cd $REPO
  python tools/pmc_summary.py gpurun_out/pmc_viz_$TAG gpurun_out/${TAG}_pmc_traffic_vizdoom.json > gpurun_out/${TAG}_pmc_traffic_vizdoom.md
  find gpurun_out/pmc_viz_$TAG -name "*.db" -size +1M -delete
fi
