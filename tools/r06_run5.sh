set -x
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -x -q -k "conv or wide or batch_norm" > gpurun_out/r06f_pytest_conv.log 2>&1; tail -3 gpurun_out/r06f_pytest_conv.log
python tools/bench_conv_wide.py > gpurun_out/r06f_bench_conv_wide.log 2>&1; cat gpurun_out/r06f_bench_conv_wide.log
