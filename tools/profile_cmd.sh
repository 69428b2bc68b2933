#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel trace of an arbitrary python command.
# usage: tools/profile_cmd.sh <tag> <script> [args...]  -> gpurun_out/prof_<tag>/
TAG=$1; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
SCRIPT=$REPO/$1; shift
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python $SCRIPT "$@" > $OUT/stdout.log 2> $OUT/stderr.log
DB=$(find $OUT -name "*.db" | head -1)
python $REPO/tools/rocpd_by_grid.py $DB > $OUT/by_grid.md
python $REPO/tools/rocpd_summary.py $DB > $OUT/summary.md
