set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r06l_pytest.log 2>&1; tail -3 gpurun_out/r06l_pytest.log
python tools/bench_conv2_dgrad.py > gpurun_out/r06l_bench_conv2_dgrad.log 2>&1; cat gpurun_out/r06l_bench_conv2_dgrad.log
