#!/bin/bash
# Run on the GPU box (through gpurun): MFMA-pipe busy cycles per kernel for the default bench
# (MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * SIMDs), rocprofv3 -L).  Own pass,
# --kernel-trace only.  usage: tools/profile_mfma.sh <tag>   [BENCH_ARGS=...]
set -x
TAG=${1:-r01}
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export D2P_GRAPH=0 D2P_SIDE_STREAM=0
REPO=$PWD
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT -o pmc_MFMA -- python $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline $BENCH_ARGS > $OUT/MFMA_stdout.log 2> $OUT/MFMA_stderr.log
ls -la $OUT
