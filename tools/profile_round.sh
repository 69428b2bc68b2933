set -x
python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
python bench.py > gpurun_out/r01t_bench.json 2> gpurun_out/r01t_bench.err
D2P_NO_GRAPH=1 D2P_NO_SIDE_STREAM=1 bash tools/profile_bench.sh r01t > /dev/null 2>&1
DB=$(find gpurun_out/prof_r01t -name "*.db" | head -1)
python tools/rocpd_summary.py $DB > gpurun_out/r01t_kernel_stats_serial.md
bash tools/profile_pmc.sh r01t > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc_r01t gpurun_out/r01t_pmc_traffic.json > gpurun_out/r01t_pmc_traffic.md
ls -la gpurun_out/ | tail
rm -rf gpurun_out/prof_r01t/*/*.db gpurun_out/pmc_r01t/*/*.db 2>/dev/null; find gpurun_out -name "*.db" -size +20M -delete
