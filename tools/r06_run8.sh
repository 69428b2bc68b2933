set -x
mkdir -p gpurun_out
python -m pytest tests/test_dp_two_ranks_gpu.py tests/test_model_gpu.py -x -q -k "two_ranks or rccl" > gpurun_out/r06h_pytest_dp.log 2>&1; tail -3 gpurun_out/r06h_pytest_dp.log
for OV in 0 1 0 1; do
D2P_DP_OVERLAP=$OV python bench.py --gpus 1 --self-spawn --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-h2d --no-config4 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('overlap $OV', d['ms_per_step'], d['value'], d['persistent_lstm_fallbacks'], d['rccl_ranks_seen'])" >> gpurun_out/r06h_dp_overlap_one_rank.log
done
cat gpurun_out/r06h_dp_overlap_one_rank.log
# config 4 kernel trace, one stream
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out/prof_r06h_viz
mkdir -p $OUT
(cd /tmp && D2P_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats -d $OUT -o bench -- python $REPO/bench.py --preset vizdoom --steps 5 --warmup 3 --no-cpu-baseline --no-roofline --no-h2d > $OUT/stdout.log 2> $OUT/stderr.log)
DB=$(find $OUT -name "*.db" | head -1)
python tools/rocpd_summary.py $DB 8 > gpurun_out/r06h_kernel_stats_vizdoom.md
find gpurun_out -name "*.db" -size +1M -delete
head -40 gpurun_out/r06h_kernel_stats_vizdoom.md | cut -c1-150
