#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel-trace + stats of the default bench.
# usage: tools_profile.sh <tag>   -> gpurun_out/prof_<tag>/  (copy summaries to profiles/)
set -x
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT -o bench -- python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-h2d --no-config4 $BENCH_ARGS > $OUT/bench_stdout.log 2> $OUT/bench_stderr.log
ls -R $OUT | head -30
