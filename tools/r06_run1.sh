set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r06a_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06a_pytest.log
python bench.py > gpurun_out/r06a_bench.json 2> gpurun_out/r06a_bench.err
bash tools/profile_bench.sh r06ag > /dev/null 2>&1
DB=$(find gpurun_out/prof_r06ag -name "*.db" | head -1)
python tools/rocpd_summary.py $DB 13 > gpurun_out/r06a_kernel_stats_default.md
python tools/rocpd_streams.py $DB 5 --seq > gpurun_out/r06a_streams_timeline.txt
find gpurun_out -name "*.db" -size +1M -delete
tail -3 gpurun_out/r06a_pytest.log
