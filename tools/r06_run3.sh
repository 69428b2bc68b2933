set -x
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -x -q -k "conv or wide or batch_norm" > gpurun_out/r06c_pytest_conv.log 2>&1; tail -3 gpurun_out/r06c_pytest_conv.log
python tools/bench_conv_wide.py > gpurun_out/r06c_bench_conv_wide.log 2>&1; cat gpurun_out/r06c_bench_conv_wide.log
python -m pytest tests/test_model_gpu.py -x -q -k "vizdoom or k25 or folded" > gpurun_out/r06c_pytest_model.log 2>&1; tail -3 gpurun_out/r06c_pytest_model.log
python bench.py --preset vizdoom --steps 20 --warmup 5 --no-cpu-baseline --no-h2d > gpurun_out/r06c_bench_vizdoom.json 2> gpurun_out/r06c_bench_vizdoom.err; tail -c 600 gpurun_out/r06c_bench_vizdoom.json
