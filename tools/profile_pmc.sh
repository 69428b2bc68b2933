#!/bin/bash
# Run on the GPU box (through gpurun): HBM traffic counters for the default bench, each counter in
# its own pass with --kernel-trace only (MI355X_MICROARCH.md: FETCH_SIZE costs 3 TCC slots,
# WRITE_SIZE 2 -- they do not fit one pass).  usage: tools/profile_pmc.sh <tag>
set -x
TAG=${1:-r01}
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export D2P_GRAPH=0 D2P_SIDE_STREAM=0
REPO=$PWD
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $OUT -o pmc_$C -- python $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-h2d --no-config4 $BENCH_ARGS > $OUT/${C}_stdout.log 2> $OUT/${C}_stderr.log
done
ls -la $OUT
